cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],1))"; }
run base
for v in 8 24 32; do RTA_W_REFILL=$v run "refill $v"; done
for v in 12 32 40; do RTA_W_QUORUM=$v run "quorum $v"; done
for v in 4 8 64; do RTA_W_WAITING=$v run "waiting $v"; done
for v in 8 16 64; do RTA_W_CHUNK=$v run "chunk $v"; done
for v in 16 24; do RTA_W_CAP=$v run "cap $v"; done
run base
