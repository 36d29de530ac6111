#!/bin/bash
# PMC passes over tools/probe_layouts.py for the production batch kernel and the wide-tree kernel:
#   tools/pmc_wide.sh <tag>      (through gpurun, from the repo root)
set -u
TAG=${1:-pmcw}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export VARIANTS=${VARIANTS:-0,20} REPS=2 COPIES=${COPIES:-48}
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/a" -- python3 "$R/tools/probe_layouts.py" > "$OUT/a.log" 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/b" -- python3 "$R/tools/probe_layouts.py" > "$OUT/b.log" 2>&1 || exit 3
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/" + sub + "/*/*_counter_collection.csv")[0])):
        k = r["Kernel_Name"]
        k = "k_wide_batch" if "k_wide_batch" in k else "k_batch_closest" if "k_batch_closest" in k else "k_wide_redo" if "k_wide_redo" in k else None
        if not k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        print(sub, k, "launches", len(n[k]), "ms per launch %.3f" % (cyc / 2.4e6 / len(n[k])))
        for c, x in sorted(v.items()):
            print("    %-40s %.4e  per launch-cycle %.4f" % (c, x, x / max(cyc, 1)))
        if sub == "a":
            print("    valu_busy %.3f lanes %.3f waiting %.3f" % (v['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc), v['SQ_THREAD_CYCLES_VALU'] / max(v['SQ_ACTIVE_INST_VALU'] * 64, 1), v['SQ_WAIT_ANY'] / max(v['SQ_WAVE_CYCLES'], 1)))
PY
