#!/bin/bash
# One counter pass over one full-size render of the bench workload: vector and scalar instructions, waves and wave cycles per
# kernel.  tools/sq_pass.sh [tag]   (BENCH_ARGS: further bench.py arguments, e.g. --workload cornell1080p)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-sq}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/bench.py" --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --also none ${BENCH_ARGS:-} > "$OUT/pmc_sq.log" 2>&1 || exit 5
python3 - "$OUT/pmc_sq" <<'P'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:30]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], name)
    if key not in seen:
        seen.add(key); calls[name] += 1
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:7]:
    print("%-20s launches %5d  VALU %8.2f G  SALU %7.2f G  lanes %.3f  waiting %.3f  waves %.1f M" % (
        name, calls[name], c["SQ_INSTS_VALU"] / 1e9, c["SQ_INSTS_SALU"] / 1e9,
        c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"] * 64, 1), c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), c["SQ_WAVES"] / 1e6))
P
