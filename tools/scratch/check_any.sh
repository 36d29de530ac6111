set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread 2>&1 | tail -15
