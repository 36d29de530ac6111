#!/usr/bin/env python3
"""Condenses a tools/profile_round.sh output directory (gpurun_out/<tag>) into
the files kept under profiles/: the rocprofv3 kernel statistics, a per-kernel
PMC summary, and profiles/traffic.json (HBM bytes per traced ray of the
intersect kernel, which bench.py turns into roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the
bytes of wide coalesced reads, so the read side is doubled as the MI355X notes
prescribe; narrow gathers are not calibrated, which makes the doubled figure an
upper estimate for this kernel.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = ("k_intersect_odd", "k_intersect", "k_advance", "k_shadow", "k_sum_blocks", "k_resolve")


def kernel_of(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def aggregate(path):
    files = glob.glob(os.path.join(path, "vm", "*_counter_collection.csv"))
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
    with open(path) as f:
        for line in f:
            if line.startswith("{"):
                return json.loads(line)
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", tag)
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(src, "trace", "vm", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, f"{tag}_dragon1080p_kernel_stats.csv"))
    traced = bench_line(os.path.join(src, "bench_traced.log"))
    summary = {"bench_line_under_kernel_trace": traced, "pmc_runs": "bench.py --spp 128 --steps 1 --warmup 0",
               "kernels": {}}
    for run in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
        for k, v in aggregate(os.path.join(src, run)).items():
            summary["kernels"].setdefault(k, {}).update(v)
    for k, v in summary["kernels"].items():
        n = v["launches"]
        if "FETCH_SIZE" in v:
            v["hbm_read_bytes_per_launch_counter"] = v["FETCH_SIZE"] * 1024 / n
            v["hbm_read_bytes_per_launch_corrected_x2"] = 2 * v["FETCH_SIZE"] * 1024 / n
        if "WRITE_SIZE" in v:
            v["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024 / n
        if "TCC_HIT_sum" in v:
            v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        if "SQ_THREAD_CYCLES_VALU" in v:
            v["valu_lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)
            v["wave_time_waiting_on_memory"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    pmc_bench = bench_line(os.path.join(src, "pmc_fetch.log"))
    ki = summary["kernels"].get("k_intersect", {})
    if pmc_bench and "FETCH_SIZE" in ki and "WRITE_SIZE" in ki:
        r = pmc_bench["roofline"]
        rays = r["rays_per_launch"] * r["launches"]
        per_ray = (2 * ki["FETCH_SIZE"] + ki["WRITE_SIZE"]) * 1024 / rays
        traffic = {"tag": tag, "kernel": "k_intersect", "workload": pmc_bench["config"]["workload"],
                   "hbm_bytes_per_ray": per_ray, "read_bytes_per_ray_x2": 2 * ki["FETCH_SIZE"] * 1024 / rays,
                   "write_bytes_per_ray": ki["WRITE_SIZE"] * 1024 / rays,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; KiB -> bytes; "
                             "read side doubled (gfx950 FETCH_SIZE counts 64 B per 128 B request)"}
        # vector-ALU work of the whole pipeline per ray (wave instructions; one
        # takes 4 cycles of one SIMD), for the bench line's second roofline axis
        sq_bench = bench_line(os.path.join(src, "pmc_sq.log"))
        if sq_bench:
            # every query of the run, those answered without a traversal included
            sq_rays = sq_bench["value"] * 1e6 * sq_bench["ms_per_step"] * 1e-3 * sq_bench["steps"]
            traffic["valu_wave_instructions_per_ray"] = sum(
                v.get("SQ_INSTS_VALU", 0.0) for v in summary["kernels"].values()) / sq_rays
        with open(os.path.join(dst, "traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
        summary["intersect_traffic"] = traffic
    with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary.get("intersect_traffic"), indent=1))


if __name__ == "__main__":
    main()
