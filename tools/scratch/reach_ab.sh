set -e
cd $GRAFT_REPO_ROOT
for reach in 2.05 4 2.05 4; do
    RTA_WIDE_VERBOSE=1 RTA_WIDE_REACH=$reach timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | python -c "
import json,sys
for l in sys.stdin.read().strip().splitlines():
    if l.startswith('wide tree'): print(l)
    if l.startswith('{'):
        d=json.loads(l); print('reach $reach', round(d['value']), round(d['traced_mrays_per_s']), round(d['ms_per_step'],1), d['retraced_rays_per_step'])"
done
