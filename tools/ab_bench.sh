#!/bin/bash
# A/B timing of builds of libazhip.so on ONE GPU box (boxes differ by a few per cent, so builds are only ever
# compared inside one call): tools/ab_bench.sh <rounds> <a.so> <b.so> [...]   -> ms per step and per move launch.
# Build variants beforehand THROUGH THE MAKEFILE, so that its checks of the device assembly apply to them too, e.g.
#   make -C alphazero_amd/csrc -j6 OUT=../variants/knob.so EXTRA="-DSOME_KNOB=1"
# and run this on the GPU box; every bench run has its own time limit and the script stops at the first failure.
set -e
ROUNDS=$1; shift
for i in $(seq "$ROUNDS"); do
    for lib in "$@"; do
        echo -n "$(basename "$lib"): "
        AZHIP_LIBRARY="$(realpath "$lib")" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null \
            | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2), "ms per step,", round(d["roofline"]["avg_launch_ms"],2), "ms per move launch")'
    done
done
