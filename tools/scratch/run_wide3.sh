set -e
cd $GRAFT_REPO_ROOT
for rpl in 1 2 4; do
 for refill in 8 16 24; do
  echo "== rpl $rpl refill $refill"
  RTA_WIDE_RPL=$rpl RTA_WIDE_REFILL=$refill REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
 done
done
echo "== rpl 2 cap 28"
RTA_WIDE_RPL=2 RTA_WIDE_CAP=28 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
echo "== rpl 2 cap 24"
RTA_WIDE_RPL=2 RTA_WIDE_CAP=24 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep bounce
