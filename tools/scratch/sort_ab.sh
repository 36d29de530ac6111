set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for v in 0 1 0 1; do
    RTA_SORT_LANES=$v timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline']['exclusive_pass']; print('sort $v', round(d['value']), round(d['traced_mrays_per_s']), round(d['ms_per_step'],1), 'excl', round(e['intersect_ms'],1), round(e['advance_ms'],1))"
done
for v in 0 1; do
RTA_SORT_LANES=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload cornell1080p | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cornell sort $v', round(d['value']), round(d['ms_per_step'],1))"
done
