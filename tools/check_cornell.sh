#!/bin/bash
# After a change to the kernels that serve scenes without a wide tree, on the GPU box: the parity tests, then cornell 1080p
# with its retraced rays, and (FLAT_AB=1) the same with the scan of all primitives switched off.
set -e
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_cells.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for flat in ${FLAT_AB:+0} 1; do
    RTA_FLAT=$flat timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload cornell1080p | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cornell RTA_FLAT=$flat: %d Mqueries/s, %.1f ms, retraced %s' % (d['value'], d['ms_per_step'], d.get('retraced_rays_per_step')))"
done
