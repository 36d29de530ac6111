#!/bin/bash
# soup through rta_trace_closest: certified walk against the replaying traversal
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/soup_ab
mkdir -p "$OUT"
cd "$R"
timeout -k 10 300 python -u -m pytest tests/test_gpu_wide.py tests/test_gpu_parity.py -m gpu -x -q --timeout 280 --timeout-method thread > "$OUT/parity.log" 2>&1 || { tail -30 "$OUT/parity.log"; exit 1; }
tail -2 "$OUT/parity.log"
RTA_WIDE_VERBOSE=1 timeout -k 10 500 python -u tests/soup_bench.py --check 262144 --no-host-build > "$OUT/walk.log" 2>&1 || { tail -30 "$OUT/walk.log"; exit 2; }
tail -12 "$OUT/walk.log"
RTA_BATCH_EXACT=1 timeout -k 10 400 python -u tests/soup_bench.py --check 0 --no-host-build > "$OUT/exact.log" 2>&1 || { tail -30 "$OUT/exact.log"; exit 3; }
tail -8 "$OUT/exact.log"
