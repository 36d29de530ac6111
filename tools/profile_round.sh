#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box:
#   tools/profile_round.sh <tag>      (run through gpurun from the repo root)
# 1. kernel trace + stats of the default bench command,
# 2. PMC passes (separate runs, as the counter blocks do not fit together) on a
#    shorter run of the same workload: HBM read / write bytes, cache hits, SQ
#    utilisation.
# Output: gpurun_out/<tag>/...; copy the summaries you keep into profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 2 --warmup 1 > "$OUT/bench_traced.log" 2>&1 || exit 1
SHORT="--spp 128 --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" $SHORT > "$OUT/pmc_fetch.log" 2>&1 || exit 2
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" $SHORT > "$OUT/pmc_write.log" 2>&1 || exit 3
timeout -s KILL 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_l2" -- python3 "$R/bench.py" $SHORT > "$OUT/pmc_l2.log" 2>&1 || exit 4
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/bench.py" $SHORT > "$OUT/pmc_sq.log" 2>&1 || exit 5
echo done
