#!/bin/bash
# Collects the rocprofv3 evidence that profiles/ holds (run on a GPU box from the repo root):
#   tools/collect_profiles.sh <outdir>
# Kernel-trace passes and counter passes are separate runs; every pass has its own time limit.
set -e
OUT=${1:-gpurun_out/profiles_raw}
mkdir -p "$OUT"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
BENCH="bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_net" -- python3 $BENCH > "$OUT/bench_net.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_net_stepwise" -- python3 $BENCH --stepwise > "$OUT/bench_net_stepwise.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_net_cache" -- python3 $BENCH --eval-cache-log2 24 > "$OUT/bench_net_cache.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_net_fp32" -- python3 $BENCH --net-precision fp32 > "$OUT/bench_net_fp32.log" 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_step" -- python3 tools/train_step_probe.py 100 > "$OUT/train_step.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_uniform" -- python3 $BENCH --model uniform > "$OUT/bench_uniform.log" 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/net128" -- python3 tools/net128_bench.py 16384 > "$OUT/net128.log" 2>&1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $BENCH --preroll 8 > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $BENCH --preroll 8 > "$OUT/pmc_write.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc128_fetch" -- python3 tools/net128_bench.py 16384 > "$OUT/pmc128_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc128_write" -- python3 tools/net128_bench.py 16384 > "$OUT/pmc128_write.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc128_sq" -- python3 tools/net128_bench.py 16384 > "$OUT/pmc128_sq.log" 2>&1
timeout -s KILL 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 $BENCH --preroll 8 > "$OUT/pmc_sq.log" 2>&1
timeout -s KILL 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_fp32_sq" -- python3 $BENCH --preroll 8 --net-precision fp32 > "$OUT/pmc_fp32_sq.log" 2>&1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_stepwise_fetch" -- python3 $BENCH --preroll 8 --stepwise > "$OUT/pmc_stepwise_fetch.log" 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_stepwise_write" -- python3 $BENCH --preroll 8 --stepwise > "$OUT/pmc_stepwise_write.log" 2>&1
# compact summaries (the raw counter CSVs are large)
python3 tools/summarize_profiles.py "$OUT"
