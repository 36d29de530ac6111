set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in wide exact; do
  flag=""; [ $mode = exact ] && flag="--exact-only"
  for st in 1 4; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${mode}_s$st -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --streams $st $flag > $R/gpurun_out/prof_${mode}_s$st.log 2>&1
    f=$(ls $R/gpurun_out/prof_${mode}_s$st/*/*_kernel_stats.csv | head -1)
    echo "== $mode streams $st"; head -6 $f | cut -c1-200
  done
done
