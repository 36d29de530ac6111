set -u
R=$(pwd)
mkdir -p gpurun_out/r03g
timeout -k 10 600 python bench.py --steps 5 --warmup 1 > gpurun_out/r03g/bench_default.json 2> gpurun_out/r03g/bench_default.err || { tail gpurun_out/r03g/bench_default.err; exit 1; }
cut -c1-300 gpurun_out/r03g/bench_default.json
bash tools/profile_round.sh r03g || exit 2
cd "$R"
bash tools/profile_soup.sh r03g_soup || exit 3
cd "$R"
timeout -k 10 900 python -u tests/mass_check.py --rays 64 > gpurun_out/r03g/mass_check.log 2>&1 || { tail gpurun_out/r03g/mass_check.log; exit 4; }
tail -5 gpurun_out/r03g/mass_check.log
