#!/usr/bin/env python3
"""Scan gfx9 device assembly (hipcc -S --cuda-device-only) for register spills that the compiler placed in front of
the EXEC restore at the head of a basic block.

A block that ends a divergent region begins with `s_or_b64 exec, exec, sN` (the lanes that skipped the region come
back).  Spill code that the register allocator puts at the top of such a block must come AFTER that instruction:
a `scratch_store` (or an AGPR copy) that runs before it saves only the lanes that went through the region, and the
reload -- which runs with all lanes -- hands the others whatever the scratch slot held before.  This image's compiler
does place spills there in some register-starved builds.  That is what made builds of k_c4_move_fused with a
168-register budget return wrong trees and, with more games, fault: thirteen live values of the network phase were
stored at the join behind a search step, before EXEC was restored, so the lane groups that had no game in that step
lost them (tools/exp_spill_probe.py reproduces it; the same source without the out-of-line call that forced those
values out of their registers compiles without the pattern and runs correctly).

    check_spill_exec.py file.s [...]      exit status 1 and one line per finding if the pattern occurs

The build runs it over the device assembly of every source file (alphazero_amd/csrc/Makefile)."""
import re
import sys

LABEL = re.compile(r'^(\.LBB\d+_\d+|[A-Za-z_][\w$.]*):')
FALLTHROUGH = re.compile(r'^; (%bb\.\d+):')
FUNC = re.compile(r'^([A-Za-z_][\w$.]*):')
EXEC_RESTORE = re.compile(r'^\s*s_or_b64\s+exec,\s*exec,')
# spill traffic: scratch / private-buffer accesses the allocator marks, and VGPR <-> AGPR spill copies
SPILL = re.compile(r'^\s*(scratch_(store|load)\w*|buffer_(store|load)\w*)\b.*;.*(Spill|Reload)|'
                   r'^\s*v_accvgpr_(write|read)\w*\b.*;.*(Spill|Reload)')
TERMINATOR = re.compile(r'^\s*(s_branch|s_cbranch_\w+|s_endpgm|s_setpc_b64|s_swappc_b64)\b')


def scan(path):
    findings = []
    func = '?'
    block, block_line = None, 0
    pending = []              # spill instructions seen in the current block before any EXEC restore
    restored = False
    with open(path) as f:
        for n, line in enumerate(f, 1):
            code = line.split(';')[0].strip()
            fall = FALLTHROUGH.match(line)
            if fall:                                   # a block entered by falling through has no label of its own
                block, block_line, pending, restored = fall.group(1), n, [], False
                continue
            m = LABEL.match(line)
            if m:
                if FUNC.match(line) and not line.startswith('.L'):
                    func = m.group(1)
                block, block_line, pending, restored = m.group(1), n, [], False
                continue
            if not code or code.startswith('.'):
                continue
            if TERMINATOR.match(line):
                pending, restored = [], True          # the rest of the text up to the next label is a new block
                continue
            if EXEC_RESTORE.match(line):
                if pending and not restored:
                    for pn, pl in pending:
                        findings.append((path, func, block, pn, pl, n))
                pending, restored = [], True
                continue
            if not restored and SPILL.match(line):
                pending.append((n, line.strip()))
    return findings


def main(argv):
    bad = []
    for p in argv[1:]:
        bad += scan(p)
    for path, func, block, n, text, at in bad:
        print(f'{path}:{n}: in {func}, block {block}: "{text}" runs before the EXEC restore at line {at}')
    if bad:
        print(f'{len(bad)} spill instruction(s) ahead of an EXEC restore: lanes that skipped the region lose these values',
              file=sys.stderr)
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
