#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... : the default bench workload in short runs with VAR set to each value
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
VAR=$1; shift
for v in "$@"; do
    export $VAR=$v
    timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --also none ${BENCH_ARGS:-} | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v: %.1f ms, %d Mqueries/s' % (d['ms_per_step'], d['value']))" || exit 1
done
