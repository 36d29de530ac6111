set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/split
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RTA_SPLIT_INTERSECT=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace1" -- python3 "$R/bench.py" --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --also none > "$OUT/bench.log" 2>&1 || exit 1
find "$OUT/trace1" -name '*kernel_stats.csv' -exec cat {} \; | cut -c1-300
