cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],1))"; }
run
for s in 2 3 6 8; do run --streams $s; done
for n in 8388608 33554432; do run --slots $n; done
run --slots 33554432 --streams 8
for sps in 4 16 32; do run --samples-per-stream $sps; done
