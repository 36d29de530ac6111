#!/bin/bash
# kernel trace of a strong-scaled rank's share (tools/probe_shard8.py): tools/trace_shard.sh [shards] [streams]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/shard${1:-8}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SHARDS=${1:-8} SPS=8 STREAMS=${2:-0}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/tools/probe_shard8.py" > "$OUT/run.log" 2>&1 || exit 1
grep ms "$OUT/run.log"
find "$OUT/trace" -name '*kernel_stats.csv' -exec cat {} \; | cut -c1-150 | head -8
