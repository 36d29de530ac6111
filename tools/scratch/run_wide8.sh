set -e
cd $GRAFT_REPO_ROOT
export COPIES=48 RTA_WIDE_RPL=8
RTA_WIDE_COUNT=1 REPS=1 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep -v amdgpu.ids | grep -v camera | head -2
for refill in 8 16; do
 for wq in 4 8 16 64; do
  echo "== refill $refill waiting $wq"
  RTA_WIDE_REFILL=$refill RTA_WIDE_WAITING=$wq REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
 done
done
RTA_WIDE_REFILL=16 RTA_WIDE_WAITING=8 RTA_WIDE_COUNT=1 REPS=1 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep -v amdgpu.ids | grep -v camera | head -2
