set -e
cd $GRAFT_REPO_ROOT
export COPIES=48 RTA_WIDE_VERBOSE=1
RTA_WIDE_COUNT=1 REPS=1 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep -v amdgpu.ids | grep -v camera | head -4
echo "== production"
REPS=5 VARIANTS=0 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
for rpl in 4 8; do
 for q in 16 24 32; do
  echo "== rpl $rpl quorum $q"
  RTA_WIDE_RPL=$rpl RTA_WIDE_QUORUM=$q REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
 done
done
echo "== rpl 8 cap 16"
RTA_WIDE_RPL=8 RTA_WIDE_CAP=16 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
echo "== rpl 8 cap 24"
RTA_WIDE_RPL=8 RTA_WIDE_CAP=24 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
echo "== rpl 8 refill 16"
RTA_WIDE_RPL=8 RTA_WIDE_REFILL=16 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
echo "== rpl 8 refill 32"
RTA_WIDE_RPL=8 RTA_WIDE_REFILL=32 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
export COPIES=8
echo "== small set production / wide rpl 4"
REPS=5 VARIANTS=0 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
RTA_WIDE_RPL=4 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
