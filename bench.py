#!/usr/bin/env python3
"""Throughput benchmark of the MI355X path-tracing backend.

    python bench.py --gpus N --steps K --warmup W

A step is one complete render of the workload (default: scenes/dragon.txt at
1920x1080, 1024 spp per GPU) through the C ABI of libraytracer_amd.so with all
inputs resident in HBM; N > 1 runs one rank per GPU under torch.distributed
(RCCL), each rank rendering its interleaved 8x8-tile shard, followed by the
all-gather of the fp32 radiance sums and the tone-mapping resolve.  Scaling is
weak: samples per pixel grow with N so that per-GPU work stays fixed.

Rank 0 prints ONE JSON line: rays/s (closest-hit + occlusion queries) over the
whole job, the live HIP-event roofline of the intersect kernel and, at N = 1, a
host-CPU baseline timed on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import raytracer_amd as rta  # noqa: E402

# Algorithmic bytes per closest-hit query: 64 B per inner-node visit (two child
# boxes), 36 B per triangle test, 16 B per sphere test, 40 B ray in + hit out,
# with the reference traversal's own visit counts per workload.
WORKLOADS = {
    "dragon1080p": dict(scene="dragon.txt", width=1920, height=1080, spp=1024, bytes_per_ray=691.0, early_bytes=156.0,
                        sps=16, cpu_spp=256, label="scenes/dragon.txt 1920x1080 1024spp"),
    "cornell1080p": dict(scene="cornell_box.txt", width=1920, height=1080, spp=256, bytes_per_ray=764.0, early_bytes=104.0,
                         sps=16, cpu_spp=64, label="scenes/cornell_box.txt 1920x1080 256spp"),
    "dragon4k": dict(scene="dragon.txt", width=3840, height=2160, spp=4096, bytes_per_ray=691.0, early_bytes=156.0,
                     sps=32, cpu_spp=64, label="scenes/dragon.txt 3840x2160 4096spp"),
}
# sps: samples per random stream.  A stream's samples are a sequential chain (one wavefront step per
# path segment), so shorter streams mean more chains in flight and a shorter end of the render, at
# 16 bytes of block sum per stream: 1080p x 1024 spp runs at 12.3 / 12.3 / 11.5 / 10.2 Grays/s with
# 8 / 16 / 32 / 64 samples per stream; the 4K frame has streams enough either way (13.4 / 13.7).
# early_bytes: what the reference traversal moves for a closest-hit query that the GPU path answers
# without a traversal (a camera ray that misses every leaf's box): 40 B ray and hit + 64 B per node
# visit.  One visit on cornell (its root has two leaves); 1.8 on average on dragon = the reference's
# 651 B of node and triangle reads per query less the 567 B per query that `--count-work` counts on
# the device for the rays that are traced, over the 72 % of the queries that are not.
HBM_PEAK_GBS = 8000.0


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
           "sample": f"{w['scene']} {w['width']}x{w['height']} at {w['cpu_spp']} spp, row-seeded, "
                     f"{rays} rays, OpenMP rows dynamic", "seconds": port_s,
           "msamples_per_s": cn["samples"] / port_s / 1e6}
    if os.path.exists(orc.REF_BIN):
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "scene.txt")
            with open(path, "w") as f:
                f.write(text)
            t0 = time.perf_counter()
            rc = subprocess.run([orc.REF_BIN, path], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            ref_s = time.perf_counter() - t0
            if rc.returncode == 0:
                out.update({"value": rays / ref_s / 1e6, "kind": "reference", "seconds": ref_s,
                            "msamples_per_s": cn["samples"] / ref_s / 1e6,
                            "port_mrays_per_s": rays / port_s / 1e6})
                out["sample"] += "; whole-process wall time of the reference binary (parse+BVH+BMP included)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel per GPU")
    ap.add_argument("--samples-per-stream", type=int, default=0, help="0: the workload's own (16 or 32)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--run-blocks", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--count-work", action="store_true", help="also count node visits / primitive tests (slower)")
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
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    w = dict(WORKLOADS[args.workload])
    args.samples_per_stream = args.samples_per_stream or w["sps"]
    spp_per_gpu = args.spp or w["spp"]
    spp = spp_per_gpu * world              # weak scaling: per-GPU samples fixed
    width, height = w["width"], w["height"]

    scene = rta.Scene.load(os.path.join(ROOT, "scenes", w["scene"])).build_bvh()
    scene.set_render(width, height, spp)
    ctx = rta.Context(local)
    ctx.upload(scene)
    cam = scene.camera()
    s = scene.settings

    # outputs live in HBM: the shard's radiance sums land in a frame-sized buffer
    from raytracer_amd.dist import ShardGather
    frame = torch.zeros((height, width, 3), dtype=torch.float32, device=dev)
    bgr = torch.zeros((height, width, 3), dtype=torch.uint8, device=dev)
    gatherer = ShardGather(width, height, rank, world, dev)

    p = rta.RenderParams(width=width, height=height, samples=spp, exposure=s.exposure, seed=69420,
                         samples_per_stream=args.samples_per_stream, shard_index=rank, shard_count=world,
                         slots=args.slots, out_mem=rta.MEM_DEVICE, run_blocks=args.run_blocks, streams=args.streams,
                         flags=rta.RENDER_TIME_KERNELS | (rta.RENDER_COUNT_WORK if args.count_work else 0))
    p.sky[:] = list(s.sky)

    def step():
        st = ctx.render_device(cam, p, frame.data_ptr(), None)
        if world > 1:
            # radiance gather over xGMI: one all-gather of equal-sized (padded) shards
            gatherer.gather(frame, dist)
            torch.cuda.current_stream().synchronize()     # the resolve runs on the library's own stream
        ctx.resolve_device(frame.data_ptr(), width, height, spp, s.exposure, bgr.data_ptr())
        return st

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0

    tot = {k: sum(st[k] for st in stats) for k in stats[0]}
    local_v = torch.tensor([elapsed, float(tot["rays"] + tot["shadow_rays"]), float(tot["samples"]),
                            tot["intersect_ms"], float(tot["rays"]), float(tot["intersect_launches"]), tot["gpu_ms"]],
                           dtype=torch.float64, device=dev)
    if dist is not None:
        mx = local_v.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = local_v.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        rays_all, samples_all = float(sm[1]), float(sm[2])
    else:
        rays_all, samples_all = float(local_v[1]), float(local_v[2])

    if rank == 0:
        # dominant kernel: closest-hit intersect, rank 0's launches
        launches = max(tot["intersect_launches"], 1)
        avg_ms = tot["intersect_ms"] / launches
        # bytes_per_ray averages over every closest-hit query of the workload; a query answered
        # where its ray is generated never reaches this kernel, and its share of that total
        # (early_bytes) is not credited to it
        traced = tot["rays"] - tot["early_rays"]
        rays_per_launch = traced / launches
        kernel_bytes = tot["rays"] * w["bytes_per_ray"] - tot["early_rays"] * w["early_bytes"]
        traced_bytes_per_ray = kernel_bytes / max(traced, 1)
        achieved = kernel_bytes / launches / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        line = {
            "metric": "Mrays/s + samples/s (whole node) on dragon.txt 1080p at 1/2/4/8 MI355X",
            "value": rays_all / elapsed / 1e6,
            "unit": "Mrays/s",
            "samples_per_s": samples_all / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic camera rays over the reference's shipped scene file (scenes/%s), per-(pixel,block) "
                    "seeded streams" % w["scene"],
            "config": {"workload": w["label"] if world == 1 else
                       "%s %dx%d, %d spp per GPU (%d total), 8x8 tiles interleaved over %d GPUs, RCCL all-gather of "
                       "fp32 radiance" % (w["scene"], width, height, spp_per_gpu, spp, world),
                       "samples_per_stream": args.samples_per_stream, "rays_per_sample": rays_all / samples_all},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_intersect", "avg_launch_ms": avg_ms, "launches": launches,
                         "rays_per_launch": rays_per_launch, "bytes_per_ray": traced_bytes_per_ray,
                         "bytes_per_query_all": w["bytes_per_ray"],
                         "queries_answered_early": tot["early_rays"] / max(tot["rays"], 1),
                         "concurrent_streams": args.streams or 4,
                         "pipeline_frac": rays_all / elapsed * w["bytes_per_ray"] / 1e9 / HBM_PEAK_GBS / world,
                         "intersect_launch_time_over_gpu_time": tot["intersect_ms"] / max(tot["gpu_ms"], 1e-9)},
        }
        # HBM bytes per intersect launch from the committed PMC profile of this workload
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("workload") == w["label"] and tj.get("kernel") == "k_intersect":
                line["roofline"]["traffic"] = tj["hbm_bytes_per_ray"] * rays_per_launch
                line["roofline"]["traffic_source"] = "profiles/traffic.json (%s)" % tj.get("tag")
                if tj.get("valu_wave_instructions_per_ray"):
                    # second axis: share of the vector-ALU issue slots the whole pipeline uses
                    # (profiled wave instructions per ray x 4 cycles, 1024 SIMDs at 2.4 GHz)
                    line["roofline"]["valu_busy_frac"] = (tj["valu_wave_instructions_per_ray"] * 4.0 * rays_all /
                                                          elapsed / world / (1024 * 2.4e9))
                    line["roofline"]["valu_busy_note"] = ("profiled wave instructions x 4 cycles / SIMD cycles; above 1 "
                                                          "when sparsely populated instructions issue in fewer cycles")
        if args.count_work and tot["rays"]:
            line["roofline"]["measured_bytes_per_ray"] = (64.0 * tot["inner_visits"] + 36.0 * tot["tri_tests"] +
                                                          16.0 * tot["sph_tests"]) / (tot["rays"] +
                                                                                      tot["shadow_rays"]) + 40.0
        if world == 1 and not args.no_cpu_baseline:
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
            line["cpu_baseline"] = cpu_baseline(w, threads)
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
