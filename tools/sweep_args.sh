#!/bin/bash
# tools/sweep_args.sh "<bench args>" "<bench args>" ... : the default bench workload in short runs, one per argument string
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for v in "$@"; do
    timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --also none $v | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v: %.1f ms, %d Mqueries/s' % (d['ms_per_step'], d['value']))" || exit 1
done
