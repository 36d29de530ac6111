#!/usr/bin/env python3
"""Condenses a tools/profile_soup.sh output directory (gpurun_out/<tag>) into

  profiles/<tag>_kernel_stats.csv      rocprofv3 --stats of `bench.py --workload soup10m`
  profiles/<tag>_pmc_summary.json      raw counter sums per kernel launch + the bench lines of the runs
  profiles/pmc_roofline_soup.json      what bench.py reads: per launch of the walk's kernel on each
                                       batch, HBM bytes, L2 hit rate, vector-ALU busy share

Each PMC run traces ONE step: the first k_batch_walk dispatch is the coherent batch, the second the
incoherent one.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests
at 64 bytes, so the read side is doubled (calibrated for wide coalesced reads; the walk's scattered
64-byte node reads make the doubled figure an upper estimate, the plain one a lower).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

XCDS, SIMDS, CLOCK = 8, 1024, 2.4e9
KERNELS = ("k_batch_walk", "k_batch_closest", "k_ray_keys", "radix", "onesweep", "histogram")


def kernel_of(name):
    for k in KERNELS:
        if k in name:
            return "radix_sort" if k in ("radix", "onesweep", "histogram") else k
    return None


def per_dispatch(path):
    files = glob.glob(os.path.join(path, "*", "*_counter_collection.csv"))
    if not files:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(files[0])):
        k = kernel_of(r["Kernel_Name"])
        if k:
            acc[(k, int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
    out = collections.defaultdict(dict)
    order = collections.defaultdict(list)
    for (k, d) in sorted(acc):
        order[k].append(d)
    for k, ds in order.items():
        if k in ("k_batch_walk", "k_batch_closest", "k_ray_keys") and len(ds) == 2:
            for tag, d in zip(("coherent", "incoherent"), ds):
                out[f"{k}:{tag}"] = dict(acc[(k, d)], launches=1)
        else:
            tot = collections.defaultdict(float)
            for d in ds:
                for c, v in acc[(k, d)].items():
                    tot[c] += v
            out[k] = dict(tot, launches=len(ds))
    return out


def bench_line(path):
    if not os.path.exists(path):
        return None
    with open(path) as f:
        for line in f:
            if line.startswith("{"):
                return json.loads(line)
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03_soup"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
    summary = {"tag": tag, "runs": {}, "kernels": {}}
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    summary["runs"]["kernel_trace"] = bench_line(os.path.join(src, "bench_traced.log"))
    for run in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_mem"):
        summary["runs"][run] = bench_line(os.path.join(src, run + ".log"))
        for k, v in per_dispatch(os.path.join(src, run)).items():
            summary["kernels"].setdefault(k, {}).update(v)
    with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    ref = summary["runs"].get("kernel_trace") or summary["runs"].get("pmc_fetch")
    if not ref:
        print("no runs found")
        return
    rays = ref["batches"]["incoherent"]["rays"]
    out = {"tag": tag, "workload": ref["config"]["workload"], "rays_per_batch": rays,
           "pmc_runs": "bench.py --workload soup10m --steps 1 --warmup 0 --no-cpu-baseline --no-roofline, one run per counter set",
           "formulas": {
               "hbm_bytes_per_launch": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (read side doubled: gfx950 FETCH_SIZE tallies 128-B "
                                       "requests at 64 B); hbm_bytes_per_launch_undoubled: (FETCH_SIZE + WRITE_SIZE) x 1024",
               "kernel_ms_from_counters": "GRBM_GUI_ACTIVE / 8 XCDs / 2.4 GHz",
               "valu_busy": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
               "valu_lane_utilisation": "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)",
               "waves_waiting": "SQ_WAIT_ANY / SQ_WAVE_CYCLES",
               "l2_hit_rate": "TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)",
               "tcp_hit_rate": "1 - TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum (vector-L1 accesses that did not go to L2)"},
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
            e["hbm_bytes_per_launch_undoubled"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / n
            if ":" in k:
                e["hbm_bytes_per_ray"] = e["hbm_bytes_per_launch"] / rays
                e["hbm_bytes_per_ray_undoubled"] = e["hbm_bytes_per_launch_undoubled"] / rays
        if "GRBM_GUI_ACTIVE" in v:
            ms = v["GRBM_GUI_ACTIVE"] / XCDS / CLOCK * 1e3
            e["kernel_ms_from_counters"] = ms / n
            if "hbm_bytes_per_launch" in e and ms > 0:
                e["hbm_gbs"] = e["hbm_bytes_per_launch"] * n / (ms * 1e-3) / 1e9
                e["hbm_frac_of_8tbs"] = e["hbm_gbs"] / 8000.0
                e["hbm_frac_of_6300gbs_achievable"] = e["hbm_gbs"] / 6300.0
                e["hbm_gbs_undoubled"] = e["hbm_bytes_per_launch_undoubled"] * n / (ms * 1e-3) / 1e9
            if "TCP_TCC_READ_REQ_sum" in v and ms > 0:
                e["l2_read_gbs"] = v["TCP_TCC_READ_REQ_sum"] * 64 / (ms * 1e-3) / 1e9
                e["tcp_hit_rate"] = 1 - v["TCP_TCC_READ_REQ_sum"] / max(v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), 1)
            if "SQ_ACTIVE_INST_VALU" in v:
                e["valu_busy"] = v["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * v["GRBM_GUI_ACTIVE"] / XCDS)
        if "TCC_HIT_sum" in v:
            e["l2_hit_rate"] = v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1)
        if "SQ_THREAD_CYCLES_VALU" in v:
            e["valu_lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)
            e["waves_waiting"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
            if ":" in k:
                e["valu_wave_instructions_per_ray"] = v["SQ_INSTS_VALU"] / rays
        out["kernels"][k] = e
    with open(os.path.join(dst, "pmc_roofline_soup.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out["kernels"].items() if k.startswith("k_batch_walk")}, indent=1))


if __name__ == "__main__":
    main()
