set -e
cd $GRAFT_REPO_ROOT
export RTA_WIDE_VERBOSE=1
for lin in 0 1; do
for reach in 2.05 4; do
echo "== soup linear $lin reach $reach"
RTA_WIDE_LINEAR=$lin RTA_WIDE_REACH=$reach RTA_WIDE_COUNT=1 timeout -k 10 200 python -u tools/probe_soup.py 2>&1 | grep -v amdgpu.ids | sort -u
done
done
export RTA_WIDE_RPL=8 RTA_WIDE_REFILL=16 RTA_WIDE_WAITING=16 RTA_WIDE_SHARED=1
for lin in 0 1; do
echo "== dragon linear $lin"
RTA_WIDE_LINEAR=$lin RTA_WIDE_COUNT=1 TILE=8 REPS=1 VARIANTS=20 timeout -k 10 120 python -u tools/probe_pairs.py 2>&1 | grep "wide" | sort -u
RTA_WIDE_LINEAR=$lin TILE=8 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_pairs.py 2>&1 | grep bounce
done
