set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --workload dragon4k --steps 2 --warmup 1 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dragon4k', round(d['value']), round(d['traced_mrays_per_s']), round(d['ms_per_step'],1))"
RTA_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep '^{' | cut -c1-1200
