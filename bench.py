#!/usr/bin/env python3
"""Throughput benchmark of the MI355X path-tracing backend.

    python bench.py --gpus N --steps K --warmup W

A step is one complete render of the workload (default: scenes/dragon.txt at
1920x1080, 1024 spp per GPU) through the C ABI of libraytracer_amd.so with all
inputs resident in HBM; N > 1 runs one rank per GPU under torch.distributed
(RCCL), each rank rendering its interleaved 8x8-tile shard, followed by the
all-gather of the fp32 radiance sums and the tone-mapping resolve.

Rank 0 prints ONE JSON line.

  value             closest-hit + occlusion QUERIES per second over the whole
                    job, counted as the reference counts them (one per
                    NearestHit / Occluded call of its loop), so that it compares
                    with the CPU baseline; a query that the GPU path answers
                    without walking the tree still counts
  traced_mrays_per_s  the queries that did walk the tree, per second
  samples_per_s     camera samples per second, the end-to-end figure
  scaling           "weak" by default: samples per pixel grow with N, per-GPU
                    work is fixed.  For N > 1 the line also carries "strong":
                    the same K steps of the FIXED workload split over the N
                    ranks, and "gather_ms", the all-gather on its own.
  roofline          the dominant kernel (k_intersect) from an extra, untimed
                    render on ONE HIP stream with every launch bracketed by HIP
                    events (exclusive durations) and an extra render that
                    counts node visits and primitive tests on the device.
  cpu_baseline      N = 1 only: the reference's CPU path on this host.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

import raytracer_amd as rta  # noqa: E402

# survey_bytes_per_ray: algorithmic bytes per closest-hit query with the REFERENCE traversal's
# visit counts averaged over every query of the workload (64 B per inner-node visit, 36 B per
# triangle test, 16 B per sphere test, 40 B ray in + hit out).
WORKLOADS = {
    "dragon1080p": dict(scene="dragon.txt", width=1920, height=1080, spp=1024, survey_bytes_per_ray=691.0,
                        sps=8, cpu_spp=256, label="scenes/dragon.txt 1920x1080 1024spp"),
    "cornell1080p": dict(scene="cornell_box.txt", width=1920, height=1080, spp=256, survey_bytes_per_ray=764.0,
                         sps=16, cpu_spp=64, label="scenes/cornell_box.txt 1920x1080 256spp"),
    "dragon4k": dict(scene="dragon.txt", width=3840, height=2160, spp=4096, survey_bytes_per_ray=691.0,
                     sps=32, cpu_spp=64, label="scenes/dragon.txt 3840x2160 4096spp"),
}
# sps: samples per random stream.  A stream's samples are a sequential chain (one wavefront step per
# path segment), so shorter streams mean more chains in flight and a shorter end of the render, at
# 16 bytes of block sum per stream.  dragon 1080p x 1024 spp on one GPU: 208 / 201 / 197 / 200 ms with
# 2 / 4 / 8 / 16 samples per stream; one eighth of its tiles (a rank's share of the frame split over
# 8 GPUs): 32 / 32 / 34 / 42 ms.  The image depends on this number (it defines the streams), so it
# is one number per workload, not one per GPU count.
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0
SIMDS, CLOCK_HZ = 1024, 2.4e9


def host_threads():
    """Threads for the CPU baseline: OMP_NUM_THREADS where the environment sets it (the GPU box
    gives a job its share of the machine that way), else every CPU this process may run on."""
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if env > 0:
        return env
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(w, threads):
    """Times the reference's CPU path on this host for a bounded sample of the
    workload (same scene and resolution, fewer samples per pixel; the rate does
    not depend on spp).  Uses the reference binary when it was built
    (oracle/_ref/raytracer), else the checker's restatement (oracle/liboracle.so)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    text = orc.scene_variant(w["scene"], w["width"], w["height"], w["cpu_spp"])
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    # ray counts come from the restatement, whose traversal is the reference's
    sc = orc.Scene.parse(text).build_bvh()
    t0 = time.perf_counter()
    _, _, cn = sc.render(orc.RNG_ROW, orc.MATH_LIBM, threads=threads)
    port_s = time.perf_counter() - t0
    rays = cn["rays"] + cn["shadow_rays"]
    out = {"value": rays / port_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "host_cpu_count": os.cpu_count(),
           "sample": f"{w['scene']} {w['width']}x{w['height']} at {w['cpu_spp']} spp, row-seeded, "
                     f"{rays} rays, OpenMP rows dynamic", "seconds": port_s,
           "msamples_per_s": cn["samples"] / port_s / 1e6,
           "render_loop_only_mrays_per_s": rays / port_s / 1e6,
           "render_loop_only_source": "the restatement's render call (no parse, BVH build or BMP)"}
    if os.path.exists(orc.REF_BIN):
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "scene.txt")
            with open(path, "w") as f:
                f.write(text)
            # parse + BVH + BMP of the reference process, to take them out of its wall time
            with open(os.path.join(tmp, "setup.txt"), "w") as f:
                f.write(orc.scene_variant(w["scene"], 8, 8, 1))
            env = dict(os.environ, OMP_NUM_THREADS=str(threads))
            t0 = time.perf_counter()
            rc0 = subprocess.run([orc.REF_BIN, os.path.join(tmp, "setup.txt")], cwd=tmp, env=env,
                                 stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            setup_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            rc = subprocess.run([orc.REF_BIN, path], cwd=tmp, env=env, stdout=subprocess.DEVNULL,
                                stderr=subprocess.DEVNULL)
            ref_s = time.perf_counter() - t0
            if rc.returncode == 0:
                loop_s = ref_s - setup_s if rc0.returncode == 0 and setup_s < ref_s else ref_s
                out.update({"value": rays / loop_s / 1e6, "kind": "reference", "seconds": loop_s,
                            "msamples_per_s": cn["samples"] / loop_s / 1e6,
                            "whole_process_seconds": ref_s, "setup_seconds": setup_s,
                            "port_mrays_per_s": rays / port_s / 1e6})
                out["sample"] += ("; wall time of the reference binary less the wall time of the same binary on an "
                                  "8x8x1 render of the scene (process start, parse, BVH build, BMP)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel per GPU")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="what `value` measures for N > 1: weak = spp x N (per-GPU work fixed), strong = the fixed "
                         "workload split over the ranks; the other one is reported beside it")
    ap.add_argument("--sharding", default="tiles", choices=("tiles", "samples"),
                    help="how N > 1 ranks split a frame: interleaved 8x8 tiles + all-gather (exact, default) or ranges of "
                         "the sample blocks + all-reduce(sum)")
    ap.add_argument("--samples-per-stream", type=int, default=0, help="0: the workload's own (16 or 32)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--run-blocks", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--exact-only", action="store_true",
                    help="every ray through the order-replaying traversal (RTA_RENDER_EXACT_ONLY), to price the certified walk")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the two extra renders behind `roofline`")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RTA_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo, to run the N > 1 code path on a one-GPU box
    # (the numbers of such a run mean nothing)
    rehearsal = os.environ.get("RTA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    w = dict(WORKLOADS[args.workload])
    args.samples_per_stream = args.samples_per_stream or w["sps"]
    spp_fixed = args.spp or w["spp"]
    width, height = w["width"], w["height"]

    scene = rta.Scene.load(os.path.join(ROOT, "scenes", w["scene"])).build_bvh()
    scene.set_render(width, height, spp_fixed)        # the camera's image plane follows the frame size
    ctx = rta.Context(local)
    ctx.upload(scene)
    cam = scene.camera()
    s = scene.settings

    # outputs live in HBM: the shard's radiance sums land in a frame-sized buffer
    from raytracer_amd.dist import ShardGather, block_ranges, reduce_partial_sums
    frame = torch.zeros((height, width, 3), dtype=torch.float32, device=dev)
    bgr = torch.zeros((height, width, 3), dtype=torch.uint8, device=dev)
    gatherer = ShardGather(width, height, rank, world, dev)
    # the library works on its own streams: everything torch has queued for these buffers must be through
    torch.cuda.synchronize()

    by_samples = args.sharding == "samples" and world > 1

    def params(spp, flags=0, streams=None):
        if args.exact_only:
            flags |= rta.RENDER_EXACT_ONLY
        p = rta.RenderParams(width=width, height=height, samples=spp, exposure=s.exposure, seed=69420,
                             samples_per_stream=args.samples_per_stream, shard_index=rank, shard_count=world,
                             slots=args.slots, out_mem=rta.MEM_DEVICE, run_blocks=args.run_blocks,
                             streams=args.streams if streams is None else streams, flags=flags)
        if by_samples:
            # every rank renders all pixels for its range of the sample blocks
            p.shard_index, p.shard_count = 0, 1
            p.block_begin, p.block_end = block_ranges(spp, args.samples_per_stream, world)[rank]
        p.sky[:] = list(s.sky)
        return p

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(spp):
        """W warm-up steps, then K timed steps of the job at `spp` samples per pixel over all ranks."""
        p = params(spp)

        def step():
            if by_samples:
                if p.block_end > p.block_begin:
                    st = ctx.render_device(cam, p, frame.data_ptr(), None)
                else:
                    frame.zero_()
                    torch.cuda.synchronize()
                    st = dict.fromkeys(rta.Stats().as_dict(), 0)
                reduce_partial_sums(frame, dist)          # radiance reduce over xGMI: one all-reduce(sum) of the frame
                torch.cuda.current_stream().synchronize()
                ctx.resolve_device(frame.data_ptr(), width, height, spp, s.exposure, bgr.data_ptr())
                return st
            st = ctx.render_device(cam, p, frame.data_ptr(), None)
            if world > 1:
                # radiance gather over xGMI: one all-gather of equal-sized (padded) shards
                gatherer.gather(frame, dist)
                torch.cuda.current_stream().synchronize()     # the resolve runs on the library's own stream
            ctx.resolve_device(frame.data_ptr(), width, height, spp, s.exposure, bgr.data_ptr())
            return st

        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        stats = [step() for _ in range(args.steps)]
        fence()
        elapsed = time.perf_counter() - t0
        tot = {k: sum(st[k] for st in stats) for k in stats[0]}
        v = torch.tensor([elapsed, float(tot["rays"] + tot["shadow_rays"]), float(tot["samples"]),
                          float(tot["rays"] - tot["early_rays"] + tot["shadow_rays"]), tot["gpu_ms"], float(tot["retraced_rays"]),
                          float(tot["walked_rays"])],
                         dtype=torch.float64, device=dev)
        if dist is not None:
            mx = v.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = v.clone()
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            return dict(elapsed=float(mx[0]), queries=float(sm[1]), samples=float(sm[2]), traced=float(sm[3]),
                        gpu_ms_max=float(mx[4]), spp=spp, retraced=float(sm[5]), walked=float(sm[6]))
        return dict(elapsed=float(v[0]), queries=float(v[1]), samples=float(v[2]), traced=float(v[3]),
                    gpu_ms_max=float(v[4]), spp=spp, retraced=float(v[5]), walked=float(v[6]))

    spp_weak = spp_fixed * world
    main_run = run(spp_weak if args.scaling == "weak" else spp_fixed)
    other_run = None
    if world > 1:
        other_run = run(spp_fixed if args.scaling == "weak" else spp_weak)

    # the all-gather on its own (N > 1)
    gather_ms = None
    if world > 1:
        fence()
        t0 = time.perf_counter()
        for _ in range(5):
            if by_samples:
                reduce_partial_sums(frame, dist)
            else:
                gatherer.gather(frame, dist)
        fence()
        gather_ms = (time.perf_counter() - t0) / 5 * 1e3

    def summary(r):
        return {"value": r["queries"] / r["elapsed"] / 1e6, "unit": "Mrays/s",
                "traced_mrays_per_s": r["traced"] / r["elapsed"] / 1e6,
                "samples_per_s": r["samples"] / r["elapsed"], "ms_per_step": r["elapsed"] / args.steps * 1e3,
                "spp_total": r["spp"]}

    # the two extra renders behind `roofline` (rank 0's shard, after the timed region)
    excl = count = None
    if rank == 0 and not args.no_roofline:
        spp = main_run["spp"]
        excl = ctx.render_device(cam, params(spp, rta.RENDER_TIME_KERNELS, streams=1), frame.data_ptr(), None)
        count = ctx.render_device(cam, params(spp, rta.RENDER_COUNT_WORK), frame.data_ptr(), None)

    if rank == 0:
        m = summary(main_run)
        line = {
            "metric": "Mrays/s + samples/s (whole node) on dragon.txt 1080p at 1/2/4/8 MI355X",
            "value": m["value"],
            "unit": "Mrays/s",
            "value_counts": "closest-hit and occlusion queries as the reference counts them (one per NearestHit / "
                            "Occluded call); traced_mrays_per_s counts only those that walked the tree",
            "traced_mrays_per_s": m["traced_mrays_per_s"],
            "traced_fraction": main_run["traced"] / main_run["queries"],
            "retraced_rays_per_step": main_run["retraced"] / args.steps,
            "walked_rays_per_step": main_run["walked"] / args.steps,
            "traversal": ("certified walk over the wide tree (bounce and shadow rays; camera rays where the camera lies within "
                          "the tree's reach), order-replaying traversal for the rest"
                          if main_run["walked"] else "order-replaying traversal (no wide tree for this scene, or --exact-only)"),
            "samples_per_s": m["samples_per_s"],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic camera rays over the reference's shipped scene file (scenes/%s), per-(pixel,block) "
                    "seeded streams" % w["scene"],
            **({"rehearsal": "all ranks on one GPU over gloo: plumbing only"} if rehearsal else {}),
            "config": {"workload": w["label"] if world == 1 else
                       "%s %dx%d, %d spp in total, %s" % (
                           w["scene"], width, height, main_run["spp"],
                           "sample blocks split over %d GPUs, RCCL all-reduce(sum) of fp32 radiance" % world if by_samples
                           else "8x8 tiles interleaved over %d GPUs, RCCL all-gather of fp32 radiance" % world),
                       "samples_per_stream": args.samples_per_stream,
                       "rays_per_sample": main_run["queries"] / main_run["samples"]},
        }
        if other_run is not None:
            line["strong" if args.scaling == "weak" else "weak"] = summary(other_run)
            line["gather_ms"] = gather_ms
            line["gather_bytes_per_rank"] = gatherer.bytes_per_rank
        if excl is not None:
            line["roofline"] = roofline(w, excl, count, m)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, host_threads())
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def roofline(w, excl, count, m):
    """The dominant kernel against the HBM roofline the metric prescribes, with
    the resources that actually bind (L2-served bytes, vector-ALU issue) beside
    it.  `excl`: stats of a render on ONE stream with every launch timed, so a
    launch's duration is its own; `count`: stats of a render that counted node
    visits and primitive tests on the device."""
    traced = excl["rays"] - excl["early_rays"]
    launches = max(excl["intersect_launches"], 1)
    avg_ms = excl["intersect_ms"] / launches
    # algorithmic bytes of the queries that walked the tree, from the device's own counts of
    # what they visited (closest-hit and occlusion traversals share the counters)
    queries = count["rays"] - count["early_rays"] + count["shadow_rays"]
    alg_bytes = 64.0 * count["inner_visits"] + 36.0 * count["tri_tests"] + 16.0 * count["sph_tests"] + 40.0 * queries
    closest_share = (count["rays"] - count["early_rays"]) / max(queries, 1)
    counted_per_ray = alg_bytes / max(queries, 1)
    rays_per_launch = traced / launches
    seconds = avg_ms * 1e-3
    # The metric's per-query figure x the queries one launch traces / the launch's own duration.  That
    # figure averages the reference's visits over ALL queries of the workload, the cheap camera rays
    # that never reach this kernel included, so it under-prices the rays that do (see counted_*).
    achieved = w["survey_bytes_per_ray"] * rays_per_launch / seconds / 1e9 if seconds > 0 else 0.0
    counted = counted_per_ray * rays_per_launch / seconds / 1e9 if seconds > 0 else 0.0
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "kernel": "k_intersect", "avg_launch_ms": avg_ms, "launches": launches,
           "rays_per_launch": rays_per_launch,
           "bytes_per_ray": w["survey_bytes_per_ray"],
           "bytes_per_ray_source": "64 B x inner visits + 36 B x triangle tests + 40 B with the reference traversal's "
                                   "visit counts averaged over every closest-hit query of the workload",
           "timing": "HIP events around every launch of an extra render on ONE stream (nothing overlaps), untimed for "
                     "`value`",
           "frac_note": ("`achieved` prices every query at the reference traversal's algorithmic bytes; the kernel serves "
                         "them from its caches (`traffic`, `hbm_frac_from_counters`, `l2_frac`), which is how the figure "
                         "can pass the HBM peak on a scene of a few kilobytes") if achieved > HBM_PEAK_GBS else None,
           "exclusive_pass": {"gpu_ms": excl["gpu_ms"], "intersect_ms": excl["intersect_ms"],
                              "advance_ms": excl["advance_ms"], "shadow_ms": excl["shadow_ms"],
                              "advance_launches": excl["advance_launches"],
                              "traced_mrays_per_s_of_the_kernel_alone": traced / max(excl["intersect_ms"], 1e-9) / 1e3,
                              "kernel_time_over_gpu_time": (excl["intersect_ms"] + excl["advance_ms"] +
                                                            excl["shadow_ms"]) / max(excl["gpu_ms"], 1e-9)},
           # What the traced queries really visit, counted on the device in this run (closest-hit and
           # occlusion traversals share the counters).  The tree is L2-resident, so these bytes are
           # L2-served: priced against the aggregate L2 bandwidth, for the kernel alone and for the
           # timed region as a whole.
           "counted_bytes_per_ray": counted_per_ray,
           "counted_source": "64 B x inner visits + 36 B x triangle tests + 16 B x sphere tests counted on the device "
                             "(RTA_RENDER_COUNT_WORK) over the queries that walked the tree, + 40 B ray in and hit out",
           "inner_visits_per_traced_query": count["inner_visits"] / max(queries, 1),
           "tri_tests_per_traced_query": count["tri_tests"] / max(queries, 1),
           "closest_hit_share_of_traced_queries": closest_share,
           "counted_gbs": counted, "l2_peak": L2_PEAK_GBS, "l2_frac": counted / L2_PEAK_GBS,
           "pipeline_l2_frac": m["traced_mrays_per_s"] * 1e6 * counted_per_ray / 1e9 / L2_PEAK_GBS}
    per_launch = w["survey_bytes_per_ray"] * rays_per_launch
    # counter-based figures of this workload from the committed rocprofv3 PMC passes
    ppath = os.path.join(ROOT, "profiles", "pmc_roofline.json")
    if os.path.exists(ppath):
        with open(ppath) as f:
            pj = json.load(f)
        if pj.get("workload") == w["label"]:
            k = pj["kernels"].get("k_intersect", {})
            if "hbm_bytes_per_launch" in k:
                # per traced ray from the profile, times this run's rays per launch
                out["traffic"] = k["hbm_bytes_per_traced_ray"] * traced / launches
                out["traffic_over_algorithmic"] = out["traffic"] / per_launch if per_launch else None
                out["hbm_frac_from_counters"] = out["traffic"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_ms > 0 else None
            out["counters"] = {"source": "profiles/pmc_roofline.json (%s)" % pj.get("tag"), "kernels": pj["kernels"]}
    return out


if __name__ == "__main__":
    main()
