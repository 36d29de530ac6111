#!/bin/bash
# One SQ counter pass over one render of a bench workload: tools/pmc_sq.sh <tag> [bench.py flags]
set -u
TAG=${1:-sq}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline "$@" > "$OUT/pmc_sq.log" 2>&1 || exit 5
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/pmc_sq/*/*_counter_collection.csv")[0])):
    import re; m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:24]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    if cyc < 1e5: continue
    print(f"{k:28s} launches {len(n[k]):5d} ms {cyc / 2.4e6:8.2f} valu_busy {v['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc):.3f} "
          f"lanes {v['SQ_THREAD_CYCLES_VALU'] / max(v['SQ_ACTIVE_INST_VALU'] * 64, 1):.3f} waiting {v['SQ_WAIT_ANY'] / max(v['SQ_WAVE_CYCLES'], 1):.3f} "
          f"valu_insts {v['SQ_INSTS_VALU']:.3e}")
PY
