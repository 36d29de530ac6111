set -e
cd $GRAFT_REPO_ROOT
for cfg in "1 2.05 20" "1 4 20" "1 4 24" "1 6 20"; do
set -- $cfg
echo "== relative $1 reach $2 cap $3"
RTA_WIDE_RELATIVE=$1 RTA_WIDE_REACH=$2 RTA_BATCH_CAP=$3 timeout -k 10 200 python -u tests/soup_bench.py --check 0 --no-host-build 2>&1 | grep -E "^(coherent|incoherent)" | cut -c1-120
done
RTA_WIDE_REACH=4 RTA_WIDE_COUNT=1 timeout -k 10 200 python -u tools/probe_soup.py 2>&1 | grep -v amdgpu.ids
