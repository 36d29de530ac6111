set -e
cd $GRAFT_REPO_ROOT
export RTA_WIDE_VERBOSE=1
RTA_WIDE_COUNT=1 REPS=1 VARIANTS=0,20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep -v amdgpu.ids
for q in 8 24 40; do
  echo "== quorum $q"
  RTA_WIDE_QUORUM=$q REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep -v amdgpu.ids
done
for cap in 16 28; do
  echo "== cap $cap"
  RTA_WIDE_CAP=$cap REPS=5 VARIANTS=20 timeout -k 10 120 python -u tools/probe_layouts.py 2>&1 | grep -v amdgpu.ids
done
