set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export RTA_SPLIT_INTERSECT=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/split2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --streams 1 > $R/gpurun_out/split2.log 2>&1
f=$(ls $R/gpurun_out/split2/*/*_kernel_stats.csv | head -1)
head -8 $f | cut -c1-200
