set -e
cd $GRAFT_REPO_ROOT
export RTA_WIDE_RPL=8 RTA_WIDE_REFILL=16 RTA_WIDE_WAITING=16
TILE=8 REPS=5 VARIANTS=0,20 timeout -k 10 120 python -u tools/probe_pairs.py 2>&1 | grep bounce
echo "== shared lists"
RTA_WIDE_SHARED=1 TILE=8 REPS=5 VARIANTS=0,20 timeout -k 10 120 python -u tools/probe_pairs.py 2>&1 | grep bounce
RTA_WIDE_SHARED=1 TILE=2 REPS=5 VARIANTS=0,20 timeout -k 10 120 python -u tools/probe_pairs.py 2>&1 | grep bounce
