#!/bin/bash
# ms per render of the bench workload at several sample counts (what a rank's share of a
# strong-scaled frame costs): tools/spp_sweep.sh [extra bench.py flags]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spp in 16 64 128 256 1024; do
	timeout -k 10 100 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --spp $spp "$@" 2>/dev/null |
		python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('spp', $spp, 'ms', round(j['ms_per_step'],2), 'Msamples/s', round(j['samples_per_s']/1e6))" || exit 1
done
