#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box:
#   tools/profile_round.sh <tag>      (run through gpurun from the repo root)
# 1. kernel trace + stats of the default bench command (4 slot groups on 4 streams),
# 2. kernel trace + stats of the same workload on ONE stream: kernel durations are exclusive there,
#    their sum stays below the frame time,
# 3. PMC passes, each a run of its own, on one full-size render of the bench workload itself
#    (1024 spp, not an extrapolation): HBM read / write bytes, L2 hits and requests, SQ utilisation,
#    vector-memory path.
# Output: gpurun_out/<tag>/...; tools/summarize_profiles.py <tag> condenses it into profiles/.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --also none > "$OUT/bench_traced.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace1" -- python3 "$R/bench.py" --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --also none > "$OUT/bench_traced1.log" 2>&1 || exit 1
ONE="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline --also none"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" $ONE > "$OUT/pmc_fetch.log" 2>&1 || exit 2
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" $ONE > "$OUT/pmc_write.log" 2>&1 || exit 3
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_l2" -- python3 "$R/bench.py" $ONE > "$OUT/pmc_l2.log" 2>&1 || exit 4
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/bench.py" $ONE > "$OUT/pmc_sq.log" 2>&1 || exit 5
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TA_TA_BUSY_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/pmc_mem" -- python3 "$R/bench.py" $ONE > "$OUT/pmc_mem.log" 2>&1 || exit 6
echo done
