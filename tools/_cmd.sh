timeout -k 10 600 python -u -m pytest tests/test_gpu_wide.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -15 || exit 2
RTA_WIDE_VERBOSE=1 timeout -k 10 300 python tools/probe_floor.py 2>&1 | grep -v amdgpu.ids || exit 1
