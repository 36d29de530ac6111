#!/bin/bash
# After a kernel change, on the GPU box (through gpurun from the repo root): the parity tests, then
# the bench workloads in short runs - dragon 1080p on four streams and on one (with the exclusive
# kernel times of the roofline pass), cornell 1080p.
set -e
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for st in 4 1; do
    timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams $st | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); e = d['roofline']['exclusive_pass']
print('dragon streams $st: %d Mqueries/s, %d traced Mrays/s, %.1f ms, retraced %.0f; alone: intersect %.1f ms, camera cells %.1f ms, left over %.1f ms, advance %.1f ms'
      % (d['value'], d['traced_mrays_per_s'], d['ms_per_step'], d['retraced_rays_per_step'], e['intersect_ms'], e['camera_cells_ms'], e['leftover_ms'], e['advance_ms']))"
done
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload cornell1080p | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cornell: %d Mqueries/s, %.1f ms' % (d['value'], d['ms_per_step']))"
