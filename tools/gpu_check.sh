#!/bin/bash
# One GPU session: the GPU tests, the default bench line, the N > 1 path rehearsed on one GPU, the soup line
# and a strong-scaled rank's share for 2 / 4 / 8 ranks.   tools/gpu_check.sh <tag> [quick]
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
if [ "${2:-}" != "quick" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }
tail -3 "$OUT/pytest.log"
fi
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > "$OUT/bench_dragon.json" 2> "$OUT/bench_dragon.err" || { tail -20 "$OUT/bench_dragon.err"; exit 2; }
cut -c1-600 "$OUT/bench_dragon.json"
RTA_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-roofline > "$OUT/bench_rehearsal2.json" 2> "$OUT/bench_rehearsal2.err" || { tail -20 "$OUT/bench_rehearsal2.err"; exit 3; }
cut -c1-400 "$OUT/bench_rehearsal2.json"
timeout -k 10 400 python bench.py --workload soup10m --steps 2 --warmup 1 > "$OUT/bench_soup.json" 2> "$OUT/bench_soup.err" || { tail -20 "$OUT/bench_soup.err"; exit 4; }
cat "$OUT/bench_soup.json"
for n in 2 4 8; do
  echo "shard 0 of $n:" >> "$OUT/shard_share.txt"
  SHARDS=$n SPS=8 timeout -k 10 120 python tools/probe_shard8.py >> "$OUT/shard_share.txt" 2>&1 || exit 5
done
cat "$OUT/shard_share.txt"
