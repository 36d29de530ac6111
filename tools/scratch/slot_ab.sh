set -e
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
    if [ $v = 1 ]; then export RTA_SLOT_RAYS=1; else unset RTA_SLOT_RAYS; fi
    timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline']['exclusive_pass']; print('slot_rays $v', round(d['value']), round(d['traced_mrays_per_s']), round(d['ms_per_step'],1), 'excl', round(e['intersect_ms'],1), round(e['advance_ms'],1))"
done
