#!/usr/bin/env python3
"""Condenses a tools/profile_round.sh output directory (gpurun_out/<tag>) into
the files kept under profiles/:

  <tag>_kernel_stats_4streams.csv   rocprofv3 --stats of the default bench command
  <tag>_kernel_stats_1stream.csv    the same workload on one stream (exclusive kernel durations)
  <tag>_pmc_summary.json            raw counter sums per kernel + the bench lines of the profiled runs
  pmc_roofline.json                 what bench.py reads: per kernel and launch, HBM bytes, L2-served
                                    bytes, vector-ALU busy share, each with its formula

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of wide coalesced
reads, so the read side is doubled as the MI355X notes prescribe; narrow gathers are not calibrated,
which makes the doubled figure an upper estimate for the traversal kernels.  GRBM_GUI_ACTIVE and the
SQ cycle counters come summed over the 8 XCDs.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = ("k_intersect_odd", "k_intersect_leafwise", "k_intersect", "k_camera_cells", "k_advance", "k_shadow", "k_sum_blocks",
           "k_resolve", "k_fill_far", "k_part_weights", "k_cover_leaves", "k_sky_map", "k_cell_lists", "k_init_slots")
XCDS, SIMDS, CUS, CLOCK = 8, 1024, 256, 2.4e9


def kernel_of(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def aggregate(path):
    files = glob.glob(os.path.join(path, "*", "*_counter_collection.csv"))
    if not files:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(files[0])):
        k = kernel_of(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
    return {k: dict(v, launches=len(seen[k])) for k, v in acc.items()}


def bench_line(path):
    if not os.path.exists(path):
        return None
    with open(path) as f:
        for line in f:
            if line.startswith("{"):
                return json.loads(line)
    return None


def stats_rows(path):
    files = glob.glob(os.path.join(path, "*", "*_kernel_stats.csv"))
    if not files:
        return None, {}
    rows = {}
    for r in csv.DictReader(open(files[0])):
        k = kernel_of(r["Name"])
        if k:
            rows[k] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                       "avg_us": float(r["AverageNs"]) / 1e3, "percent": float(r["Percentage"])}
    return files[0], rows


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", tag)
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    summary = {"tag": tag, "runs": {}, "kernel_stats": {}, "kernels": {}}
    for sub, name in (("trace", "4streams"), ("trace1", "1stream")):
        f, rows = stats_rows(os.path.join(src, sub))
        if f:
            shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats_{name}.csv"))
            summary["kernel_stats"][name] = rows
    summary["runs"]["kernel_trace_4streams"] = bench_line(os.path.join(src, "bench_traced.log"))
    summary["runs"]["kernel_trace_1stream"] = bench_line(os.path.join(src, "bench_traced1.log"))
    for run in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_mem"):
        summary["runs"][run] = bench_line(os.path.join(src, run + ".log"))
        for k, v in aggregate(os.path.join(src, run)).items():
            summary["kernels"].setdefault(k, {}).update(v)
    pmc_cmd = "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline (one full-size render per counter pass)"
    summary["pmc_runs"] = pmc_cmd
    with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)

    # the condensed file bench.py reads
    ref = summary["runs"].get("pmc_fetch") or summary["runs"].get("pmc_sq")
    if not ref:
        print("no PMC runs found")
        return
    # every render of a profiled run is counted: the warm-up and timed steps and the cold step after them
    renders = ref["steps"] + ref["warmup"] + (1 if ref.get("cold_ms_first_step") is not None else 0)
    traced = ref["traced_mrays_per_s"] * 1e6 * ref["ms_per_step"] * 1e-3 * renders     # traced queries of the profiled run
    samples = ref["samples_per_s"] * ref["ms_per_step"] * 1e-3 * renders
    out = {"tag": tag, "workload": ref["config"]["workload"], "pmc_runs": pmc_cmd,
           "renders_per_profiled_run": renders, "traced_queries_per_run": traced, "samples_per_run": samples,
           "formulas": {
               "hbm_bytes_per_launch": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launches (read side doubled: gfx950 FETCH_SIZE "
                                       "tallies 128-B requests at 64 B)",
               "l2_request_bytes_per_launch": "TCP_TCC_READ_REQ_sum x 64 B / launches (vector-L1 read requests to L2; "
                                              "the unit of a request, 64 B, is an assumption)",
               "kernel_ms_from_counters": "GRBM_GUI_ACTIVE / 8 XCDs / 2.4 GHz (profiled kernels run one at a time)",
               "valu_busy": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): share of the vector-ALU "
                            "issue cycles in use while the kernel runs alone; <= 1 by construction",
               "valu_lane_utilisation": "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): active lanes per vector "
                                        "instruction",
               "waves_waiting": "SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of wave lifetime parked on a wait"},
           "kernels": {}}
    for k, v in summary["kernels"].items():
        n = v.get("launches", 0)
        if not n:
            continue
        e = {"launches": n}
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            e["hbm_read_bytes_per_launch_x2"] = 2 * v["FETCH_SIZE"] * 1024 / n
            e["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024 / n
            e["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / n
            e["hbm_bytes_per_traced_ray"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / traced
        if "GRBM_GUI_ACTIVE" in v:
            ms = v["GRBM_GUI_ACTIVE"] / XCDS / CLOCK * 1e3
            e["kernel_ms_from_counters"] = ms
            if "hbm_bytes_per_launch" in e and ms > 0:
                e["hbm_gbs"] = e["hbm_bytes_per_launch"] * n / (ms * 1e-3) / 1e9
                e["hbm_frac_of_8tbs"] = e["hbm_gbs"] / 8000.0
            if "TCP_TCC_READ_REQ_sum" in v and ms > 0:
                e["l2_request_bytes_per_launch"] = v["TCP_TCC_READ_REQ_sum"] * 64 / n
                e["l2_read_gbs"] = v["TCP_TCC_READ_REQ_sum"] * 64 / (ms * 1e-3) / 1e9
                e["l2_frac_of_34500gbs"] = e["l2_read_gbs"] / 34500.0
            if "SQ_ACTIVE_INST_VALU" in v:
                e["valu_busy"] = v["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * v["GRBM_GUI_ACTIVE"] / XCDS)
        if "TCC_HIT_sum" in v:
            e["l2_hit_rate"] = v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1)
        if "SQ_THREAD_CYCLES_VALU" in v:
            e["valu_lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)
            e["waves_waiting"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
            e["valu_wave_instructions_per_sample"] = v["SQ_INSTS_VALU"] / samples
        out["kernels"][k] = e
    with open(os.path.join(dst, "pmc_roofline.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out["kernels"].get(k) for k in ("k_intersect", "k_camera_cells", "k_advance")}, indent=1))


if __name__ == "__main__":
    main()
