#!/bin/bash
# A / B of the training pass: the library built from the sources of a git revision against the working tree's, timed
# alternately in ONE gpurun call (timings of different boxes differ by several per cent).
#   tools/ab_train_pass.sh [revision]        (default HEAD; run from the repo root, here, not on the GPU box)
# Builds alphazero_amd/variants/ab_ref.so from `git archive <revision>` through that revision's own Makefile (so its
# build checks apply), then runs tools/train_step_probe.py with it and with the in-tree library, twice each.
set -e
REV=${1:-HEAD}
ROOT=$(pwd)
TMP=$(mktemp -d /tmp/ab_ref.XXXXXX)
git archive "$REV" alphazero_amd/csrc include tools/check_spill_exec.py | tar -x -C "$TMP"
make -s -j6 -C "$TMP/alphazero_amd/csrc" OUT="$ROOT/alphazero_amd/variants/ab_ref.so"
make -s -j6 -C "$ROOT/alphazero_amd/csrc"
rm -rf "$TMP"
/usr/local/graft/bin/gpurun --timeout 600 -- 'for i in 1 2; do echo "reference build:" >> gpurun_out/ab_train_pass.txt && AZHIP_LIBRARY=alphazero_amd/variants/ab_ref.so timeout -k 10 100 python tools/train_step_probe.py 300 2>&1 | tail -2 >> gpurun_out/ab_train_pass.txt && echo "working tree:" >> gpurun_out/ab_train_pass.txt && timeout -k 10 100 python tools/train_step_probe.py 300 2>&1 | tail -2 >> gpurun_out/ab_train_pass.txt || exit 1; done'
cat gpurun_out/ab_train_pass.txt
rm -f gpurun_out/ab_train_pass.txt alphazero_amd/variants/ab_ref.so
