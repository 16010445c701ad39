"""CPU-side checks of the C-ABI library: it loads, and exports every symbol that
include/azhip.h declares (no compute is called without a GPU)."""
import os
import re

from alphazero_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'azhip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return set(re.findall(r'\b(az_[a-z_0-9]+)\s*\(', text))


def test_header_and_binding_agree():
    assert _declared() == set(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.az_version() >= 1


def test_error_reporting_without_gpu_work():
    lib = _lib.load()
    assert lib.az_engine_create(None, None) != 0
    assert b'null' in lib.az_last_error()
    assert lib.az_render(1, None, -1, None, None, None) != 0
    assert lib.az_render(5, None, 0, None, None, None) == 0      # n == 0: nothing to do


def test_record_struct_layout():
    import ctypes as C
    assert C.sizeof(_lib.GameRecord) == 8 + 16 + 48 + 48 + 42 * 9 * 4
    assert C.sizeof(_lib.EngineConfig) == 8 * 4 + 4 * 8 + 3 * 8
