set -e
cd $GRAFT_REPO_ROOT
export RTA_WIDE_RPL=8 RTA_WIDE_REFILL=16 RTA_WIDE_WAITING=16 RTA_WIDE_SHARED=1
for c in 64 128 256 512; do
echo "== shared lists chunk $c"
RTA_WIDE_CHUNK=$c TILE=8 REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_pairs.py 2>&1 | grep bounce
done
unset RTA_WIDE_SHARED
bash tools/scratch/ab.sh
