#!/bin/bash
# PMC passes over tools/probe_layouts.py (production traversal kernel alone on
# camera and bounce rays): where the texture-addresser / L1 path stands.
#   tools/pmc_probe.sh <tag>      (through gpurun, from the repo root)
set -u
TAG=${1:-probe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export VARIANTS=${VARIANTS:-0} REPS=2
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD --output-format csv -d "$OUT/a" -- python3 "$R/tools/probe_layouts.py" > "$OUT/a.log" 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/b" -- python3 "$R/tools/probe_layouts.py" > "$OUT/b.log" 2>&1 || exit 3
echo done
