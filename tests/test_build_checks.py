"""The build's check of the device assembly (tools/check_spill_exec.py): register spills ahead of the EXEC restore
at the head of a basic block lose the values of the lanes that skipped the divergent region."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

BAD = """
_Z6kernelv:
; %bb.0:
	s_and_saveexec_b64 s[24:25], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_add_u32_e32 v1, 1, v1
.LBB0_2:
	scratch_store_dword off, v146, off offset:52 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[24:25]
	scratch_store_dword off, v118, off offset:208 ; 4-byte Folded Spill
	s_endpgm
"""
GOOD = BAD.replace("""	scratch_store_dword off, v146, off offset:52 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[24:25]""", """	s_or_b64 exec, exec, s[24:25]
	scratch_store_dword off, v146, off offset:52 ; 4-byte Folded Spill""")
# a spill at the end of a divergent block that falls through into the join is fine: it belongs to the lanes inside
FALLTHROUGH = """
_Z6kernelv:
; %bb.0:
	s_and_saveexec_b64 s[24:25], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	scratch_store_dword off, v1, off offset:4 ; 4-byte Folded Spill
; %bb.3:
	s_or_b64 exec, exec, s[30:31]
.LBB0_2:
	s_or_b64 exec, exec, s[24:25]
	scratch_load_dword v1, off, off offset:4 ; 4-byte Folded Reload
	s_endpgm
"""


def _scan(tmp_path, text):
    import check_spill_exec
    p = tmp_path / 'k.s'
    p.write_text(text)
    return check_spill_exec.scan(str(p))


def test_spill_ahead_of_the_exec_restore_is_found(tmp_path):
    found = _scan(tmp_path, BAD)
    assert len(found) == 1 and found[0][2] == '.LBB0_2' and 'v146' in found[0][4]


def test_spills_behind_the_restore_or_inside_the_region_pass(tmp_path):
    assert _scan(tmp_path, GOOD) == []
    assert _scan(tmp_path, FALLTHROUGH) == []


def test_the_built_library_passed_the_check():
    """The assembly the in-tree library was built from (kept by the Makefile) is clean, for every source file."""
    asms = sorted(glob.glob(os.path.join(ROOT, 'alphazero_amd', 'csrc', '.asm.libazhip', '*.s')))
    if not asms:
        import pytest
        pytest.skip('no device assembly in the tree (library built elsewhere)')
    srcs = glob.glob(os.path.join(ROOT, 'alphazero_amd', 'csrc', '*.hip'))
    assert len(asms) == len(srcs) >= 8                 # one assembly file per source file of the library
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_spill_exec.py')] + asms,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
