#!/bin/bash
# kernel split of the cornell workload on one stream
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cornell1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace1" -- python3 "$R/bench.py" --workload cornell1080p --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline > "$OUT/bench.log" 2>&1 || exit 1
find "$OUT/trace1" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
head -12 "$OUT/kernel_stats.csv"
tail -1 "$OUT/bench.log"
