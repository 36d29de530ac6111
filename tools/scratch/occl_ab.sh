set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u tools/probe_occluded.py 2>&1 | grep occlusion
RTA_BATCH_EXACT=1 timeout -k 10 200 python -u tools/probe_occluded.py 2>&1 | grep occlusion
