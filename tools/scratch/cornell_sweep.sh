set -e
cd $GRAFT_REPO_ROOT
for st in 2 4 6 8; do
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload cornell1080p --streams $st | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st', round(d['value']), round(d['ms_per_step'],1))"
done
for sps in 4 8 32; do
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload cornell1080p --samples-per-stream $sps | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sps $sps', round(d['value']), round(d['ms_per_step'],1))"
done
for slots in 8388608 33554432; do
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload cornell1080p --slots $slots | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slots $slots', round(d['value']), round(d['ms_per_step'],1))"
done
