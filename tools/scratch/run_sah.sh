set -e
cd $GRAFT_REPO_ROOT
export RTA_WIDE_RPL=8 RTA_WIDE_REFILL=16 RTA_WIDE_WAITING=16 RTA_WIDE_SHARED=1 RTA_WIDE_VERBOSE=1
for m in 0 1; do
if [ $m = 1 ]; then export RTA_WIDE_MIDPOINT=1; fi
echo "== midpoint $m"
RTA_WIDE_COUNT=1 TILE=8 REPS=1 VARIANTS=20 timeout -k 10 120 python -u tools/probe_pairs.py 2>&1 | grep "wide" | sort -u
TILE=8 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_pairs.py 2>&1 | grep bounce
done
