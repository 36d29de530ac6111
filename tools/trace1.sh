#!/bin/bash
# kernel trace of the default bench workload on ONE stream (exclusive kernel durations): tools/trace1.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-trace1}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --also none ${BENCH_ARGS:-} > "$OUT/bench.log" 2>&1 || exit 1
find "$OUT/trace" -name '*kernel_stats.csv' -exec cat {} \; | cut -c1-160 | head -12
