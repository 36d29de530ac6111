#!/bin/bash
# tools/ab_env.sh VAR v1 v2 ... : dragon 1080p on one stream with the exclusive kernel times, VAR set to each value
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
VAR=$1; shift
for v in "$@"; do
    export $VAR=$v
    timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --also none ${BENCH_ARGS:-} | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); e = d['roofline']['exclusive_pass']
print('$VAR=$v: %.1f ms, %d Mqueries/s; alone: intersect %.1f ms, camera cells %.1f ms, advance %.1f ms, gpu %.1f ms' % (d['ms_per_step'], d['value'], e['intersect_ms'], e['camera_cells_ms'], e['advance_ms'], e['gpu_ms']))" || exit 1
done
