set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_wide.py tests/test_gpu_parity.py -m gpu -x -q --timeout 280 --timeout-method thread 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline | cut -c1-900
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload cornell1080p | cut -c1-900
