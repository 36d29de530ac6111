set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for st in 4 1; do
  for m in "" "--exact-only"; do
    timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --streams $st $m | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st $m', round(d['value']), round(d['traced_mrays_per_s']), round(d['ms_per_step'],1), d['retraced_rays_per_step'])"
  done
done
