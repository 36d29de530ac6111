set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread 2>&1 | tail -3
RTA_WIDE_VERBOSE=1 timeout -k 10 200 python -u tests/soup_bench.py --check 0 --no-host-build 2>&1 | grep -E "^(coherent|incoherent|wide)" | cut -c1-120
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dragon', round(d['value']), round(d['traced_mrays_per_s']), round(d['ms_per_step'],1), d['retraced_rays_per_step'])"
