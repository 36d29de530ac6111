#!/bin/bash
# rocprofv3 evidence for the intersect-only workload (bench.py --workload soup10m) on the GPU box:
#   tools/profile_soup.sh <tag>      (run through gpurun from the repo root)
# 1. kernel trace + stats of the bench command, 2. PMC passes, each a run of its own over ONE step
# (one coherent and one incoherent batch of 64 Mi rays): HBM read / write bytes, L2 hits and
# requests, SQ utilisation, vector-memory path.  tools/summarize_soup.py <tag> condenses it into profiles/.
set -u
TAG=${1:-r03_soup}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="--workload soup10m --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" $B --steps 3 --warmup 1 > "$OUT/bench_traced.log" 2>&1 || exit 1
ONE="$B --steps 1 --warmup 0 --no-roofline"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" $ONE > "$OUT/pmc_fetch.log" 2>&1 || exit 2
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" $ONE > "$OUT/pmc_write.log" 2>&1 || exit 3
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_l2" -- python3 "$R/bench.py" $ONE > "$OUT/pmc_l2.log" 2>&1 || exit 4
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/bench.py" $ONE > "$OUT/pmc_sq.log" 2>&1 || exit 5
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TA_TA_BUSY_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/pmc_mem" -- python3 "$R/bench.py" $ONE > "$OUT/pmc_mem.log" 2>&1 || exit 6
echo done
