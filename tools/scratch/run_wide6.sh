set -e
cd $GRAFT_REPO_ROOT
export COPIES=48
echo "== production"
REPS=5 VARIANTS=0 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep -v amdgpu
echo "== wide rpl 8"
RTA_WIDE_RPL=8 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "closest or occlusion or render_bit" 2>&1 | tail -3
