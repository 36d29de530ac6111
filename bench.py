#!/usr/bin/env python3
"""Throughput benchmark of the MI355X path-tracing backend.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

Workloads (BASELINE.json's configurations):

  dragon1080p   scenes/dragon.txt 1920x1080, 1024 spp (the default: the
                configuration the metric is quoted on)
  cornell1080p  scenes/cornell_box.txt 1920x1080, 256 spp
  dragon4k      scenes/dragon.txt 3840x2160, 4096 spp
  soup10m       intersect only: 10 M random triangles, 64 Mi coherent + 64 Mi
                incoherent rays resident in HBM, closest hit

A step is one complete pass over the workload through the C ABI of
libraytracer_amd.so with all inputs resident in HBM: one render of the frame,
or one rta_trace_closest call per ray batch.

N > 1 runs one rank per GPU under torch.distributed (RCCL).  Started without a
launcher (`python bench.py --gpus N`), the script starts its N ranks itself as
child processes of a parent that never touches a GPU; started under
torch.distributed.run it is one of the ranks.  The default is STRONG scaling:
the fixed workload split over the ranks - interleaved 8x8 pixel tiles, one
all-gather of the fp32 radiance sums, then the tone-mapping resolve (a ray
batch: equal slices, nothing to exchange).  `--scaling weak` makes `value` the
job whose samples per pixel grow with N instead; either way the other one is
reported beside it.

Rank 0 prints ONE JSON line.

  value             closest-hit + occlusion QUERIES per second over the whole
                    job, counted as the reference counts them (one per
                    NearestHit / Occluded call of its loop), so that it compares
                    with the CPU baseline; a query that the GPU path answers
                    without walking the tree still counts
  traced_mrays_per_s  the queries that did walk the tree, per second
  samples_per_s     camera samples per second, the end-to-end figure
  cold_ms_first_step  one step after the scene was uploaded again: nothing that
                    a render prepares per view (pixel lists, the map of samples
                    that need no ray, the work lists) is reused
  roofline          the dominant kernel from HIP events on the stream it is
                    launched on (renders: an extra, untimed render on ONE stream
                    with every launch bracketed; ray batches: the events the
                    library records around every stage of a call) and from an
                    extra pass that counts node visits and primitive tests on
                    the device
  cpu_baseline      N = 1 only: the reference's CPU path on this host.
"""
import argparse
import json
import os
import socket
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
                         sps=16, cpu_spp=64, label="scenes/cornell_box.txt 1920x1080 256spp", kernel="k_intersect_flat"),
    "dragon4k": dict(scene="dragon.txt", width=3840, height=2160, spp=4096, survey_bytes_per_ray=691.0,
                     sps=32, cpu_spp=64, label="scenes/dragon.txt 3840x2160 4096spp"),
    "soup10m": dict(tris=10_000_000, rays=64 * 1024 * 1024, cpu_rays=4 << 20,
                    label="10M-triangle random soup, 64Mi coherent + 64Mi incoherent rays, closest hit"),
}
# sps: samples per random stream.  A stream's samples are a sequential chain (one wavefront step per
# path segment), so shorter streams mean more chains in flight and a shorter end of the render, at
# 16 bytes of block sum per stream.  dragon 1080p x 1024 spp on one GPU: 208 / 201 / 197 / 200 ms with
# 2 / 4 / 8 / 16 samples per stream; one eighth of its tiles (a rank's share of the frame split over
# 8 GPUs): 32 / 32 / 34 / 42 ms.  The image depends on this number (it defines the streams), so it
# is one number per workload, not one per GPU count.
HBM_PEAK_GBS = 8000.0
HBM_ACHIEVABLE_GBS = 6300.0
L2_PEAK_GBS = 34500.0
SIMDS, CLOCK_HZ = 1024, 2.4e9
METRIC = "Mrays/s + samples/s (whole node) on dragon.txt 1080p at 1/2/4/8 MI355X"


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


def start_ranks(n):
    """`--gpus N` without a launcher: this process, which has not touched a GPU and will not, starts
    the N ranks as children (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) with
    its own arguments; rank 0's JSON line goes to the inherited stdout.  Returns the exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def cpu_baseline_render(w, threads):
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


def cpu_baseline_soup(soup, batches, n_check, threads):
    """The checker's restatement of NearestHit (the reference binary stops at 16384 objects) on this
    host's threads, over a strided subset of each ray batch; also says whether the GPU's hits on
    that subset equal the checker's bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    checker = orc.Scene.from_triangles(soup).build_bvh()
    build_s = time.perf_counter() - t0
    seconds, rays, equal = 0.0, 0, True
    with ThreadPoolExecutor(threads) as pool:
        for name, (sub, got, _) in batches.items():
            pieces = np.array_split(np.arange(len(sub)), threads)
            t0 = time.perf_counter()
            want = list(pool.map(lambda ix: checker.trace_closest(sub[ix]), pieces))
            seconds += time.perf_counter() - t0
            rays += len(sub)
            for col in range(4):
                ref = np.concatenate([x[col] for x in want])
                if col == 0:
                    equal = equal and np.array_equal(got[:, 0].view(np.int32), ref)
                    hit = ref >= 0
                elif col == 1:
                    equal = equal and np.array_equal(got[:, 1].view(np.uint32), ref.view(np.uint32))
                else:
                    equal = equal and np.array_equal(got[hit, col].view(np.uint32), ref[hit].view(np.uint32))
    return {"value": rays / seconds / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "host_cpu_count": os.cpu_count(), "seconds": seconds, "tree_build_seconds": build_s,
            "sample": f"every {batches['coherent'][2]}th ray of each batch ({rays} rays in all) through the checker's "
                      f"restatement of NearestHit on {threads} threads; the reference binary caps at 16384 objects",
            "gpu_hits_equal_the_checkers": bool(equal)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel of the fixed workload")
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="what `value` measures for N > 1: strong = the fixed workload split over the ranks (default), "
                         "weak = spp x N (per-GPU work fixed); the other one is reported beside it")
    ap.add_argument("--sharding", default="tiles", choices=("tiles", "samples"),
                    help="how N > 1 ranks split a frame: interleaved 8x8 tiles + all-gather (exact, default) or ranges of "
                         "the sample blocks + all-reduce(sum)")
    ap.add_argument("--samples-per-stream", type=int, default=0, help="0: the workload's own (8, 16 or 32)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--run-blocks", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--tris", type=int, default=0, help="soup10m: triangles (default 10 M)")
    ap.add_argument("--rays", type=int, default=0, help="soup10m: rays per batch (default 64 Mi)")
    ap.add_argument("--exact-only", action="store_true",
                    help="every ray through the order-replaying traversal, to price the certified walk")
    ap.add_argument("--also", default=None,
                    help="comma-separated workloads to run after the main one, each for 2 steps, condensed under "
                         "`other_workloads` (default for the default workload on one GPU: cornell1080p,soup10m,dragon4k; "
                         "'none' switches it off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the extra passes behind `roofline`")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(start_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    env = dict(rank=rank, world=world, local=local, dev=dev, dist=dist, rehearsal=rehearsal)
    line = bench_soup(args, env) if args.workload == "soup10m" else bench_render(args, env)
    # BASELINE.json's other single-GPU configurations, beside the one the metric is quoted on
    also = args.also if args.also is not None else ("cornell1080p,soup10m,dragon4k" if args.workload == "dragon1080p" and world == 1
                                                    and not args.spp else "none")
    if world == 1 and also != "none":
        line["other_workloads"] = {}
        for name in also.split(","):
            side = argparse.Namespace(**vars(args))
            side.workload, side.steps, side.warmup, side.no_roofline, side.no_cpu_baseline = name, 2, 1, True, True
            side.samples_per_stream = side.spp = side.tris = side.rays = 0
            full = bench_soup(side, env) if name == "soup10m" else bench_render(side, env)
            keep = ("value", "unit", "ms_per_step", "steps", "warmup", "samples_per_s", "traced_mrays_per_s", "cold_ms_first_step",
                    "batches", "scene", "config")
            line["other_workloads"][name] = {k: full[k] for k in keep if k in full}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def fence(dist):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def over_ranks(values, dev, dist):
    """(max over ranks, sum over ranks, [every rank's first value]) of a list of numbers."""
    v = torch.tensor(values, dtype=torch.float64, device=dev)
    if dist is None:
        return v.tolist(), v.tolist(), [float(v[0])]
    mx, sm = v.clone(), v.clone()
    firsts = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(dist.get_world_size())]
    if dist.get_backend() == "gloo":
        mx, sm, firsts = mx.cpu(), sm.cpu(), [f.cpu() for f in firsts]
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    dist.all_gather(firsts, v[:1].to(firsts[0].device))
    return mx.tolist(), sm.tolist(), [float(f[0]) for f in firsts]


def base_line(args, env, w):
    return {
        "metric": METRIC,
        "value": None,
        "unit": "Mrays/s",
        "n_gpus": env["world"],
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": None,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        **({"rehearsal": "all ranks on one GPU over gloo: plumbing only"} if env["rehearsal"] else {}),
    }


# ------------------------------------------------------------------ frames

def bench_render(args, env):
    rank, world, local, dev, dist = env["rank"], env["world"], env["local"], env["dev"], env["dist"]
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
    from raytracer_amd.dist import ShardGather, block_ranges, reduce_partial_sums, render_share
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
            p.shard_index, p.shard_count = 0, 1      # every rank renders all pixels for its share of the sample blocks
        p.sky[:] = list(s.sky)
        return p

    def make_step(spp):
        p = params(spp)
        share = block_ranges(spp, args.samples_per_stream, world)[rank] if by_samples else None

        def step():
            if by_samples:
                st = render_share(ctx, cam, p, share, frame)
                reduce_partial_sums(frame, dist)          # radiance reduce over xGMI: one all-reduce(sum) of the frame
                torch.cuda.current_stream().synchronize()
            else:
                st = ctx.render_device(cam, p, frame.data_ptr(), None)
                if world > 1:
                    # radiance gather over xGMI: one all-gather of equal-sized (padded) shards
                    gatherer.gather(frame, dist)
                    torch.cuda.current_stream().synchronize()     # the resolve runs on the library's own stream
            ctx.resolve_device(frame.data_ptr(), width, height, spp, s.exposure, bgr.data_ptr())
            return st
        return step

    def run(spp):
        """W warm-up steps, then K timed steps of the job at `spp` samples per pixel over all ranks."""
        step = make_step(spp)
        for _ in range(args.warmup):
            step()
        fence(dist)
        t0 = time.perf_counter()
        stats = [step() for _ in range(args.steps)]
        fence(dist)
        elapsed = time.perf_counter() - t0
        tot = {k: sum(st[k] for st in stats) for k in stats[0]}
        mx, sm, per_rank = over_ranks(
            [tot["gpu_ms"] / max(args.steps, 1), elapsed, float(tot["rays"] + tot["shadow_rays"]), float(tot["samples"]),
             float(tot["rays"] - tot["early_rays"] + tot["shadow_rays"]), float(tot["retraced_rays"]), float(tot["walked_rays"]),
             float(tot["scanned_rays"])],
            dev, dist)
        return dict(elapsed=mx[1], queries=sm[2], samples=sm[3], traced=sm[4], spp=spp, retraced=sm[5], walked=sm[6],
                    scanned=sm[7], rank_render_ms=per_rank)

    spp_weak = spp_fixed * world
    main_run = run(spp_fixed if args.scaling == "strong" else spp_weak)
    other_run = None
    if world > 1:
        other_run = run(spp_weak if args.scaling == "strong" else spp_fixed)

    # One step with nothing prepared: uploading the scene again drops what the renders above left for
    # their successors (the buffers themselves stay allocated).
    ctx.upload(scene)
    cold = make_step(main_run["spp"])
    fence(dist)
    t0 = time.perf_counter()
    cold()
    fence(dist)
    cold_ms = over_ranks([(time.perf_counter() - t0) * 1e3], dev, dist)[0][0]

    # the exchange on its own (N > 1)
    gather_ms = None
    if world > 1:
        fence(dist)
        t0 = time.perf_counter()
        for _ in range(5):
            if by_samples:
                reduce_partial_sums(frame, dist)
            else:
                gatherer.gather(frame, dist)
        fence(dist)
        gather_ms = (time.perf_counter() - t0) / 5 * 1e3

    def summary(r):
        return {"value": r["queries"] / r["elapsed"] / 1e6, "unit": "Mrays/s",
                "traced_mrays_per_s": r["traced"] / r["elapsed"] / 1e6,
                "samples_per_s": r["samples"] / r["elapsed"], "ms_per_step": r["elapsed"] / args.steps * 1e3,
                "spp_total": r["spp"], "render_ms_per_rank": r["rank_render_ms"]}

    # the two extra renders behind `roofline` (rank 0's shard, after the timed region)
    excl = count = None
    if rank == 0 and not args.no_roofline:
        spp = main_run["spp"]
        excl = ctx.render_device(cam, params(spp, rta.RENDER_TIME_KERNELS, streams=1), frame.data_ptr(), None)
        count = ctx.render_device(cam, params(spp, rta.RENDER_COUNT_WORK), frame.data_ptr(), None)

    line = None
    if rank == 0:
        m = summary(main_run)
        split = ("sample blocks split over %d GPUs, RCCL all-reduce(sum) of fp32 radiance" % world if by_samples
                 else "8x8 tiles interleaved over %d GPUs, RCCL all-gather of fp32 radiance" % world)
        if world == 1:
            label = w["label"]
        elif args.scaling == "strong":
            label = "%s, %s" % (w["label"], split)
        else:
            label = "%s %dx%d, %d spp in total (%d per GPU), %s" % (w["scene"], width, height, main_run["spp"], spp_fixed, split)
        line = base_line(args, env, w)
        line.update({
            "value": m["value"],
            "value_counts": "closest-hit and occlusion queries as the reference counts them (one per NearestHit / "
                            "Occluded call); traced_mrays_per_s counts only those that walked the tree",
            "traced_mrays_per_s": m["traced_mrays_per_s"],
            "traced_fraction": main_run["traced"] / main_run["queries"],
            "retraced_rays_per_step": main_run["retraced"] / args.steps,
            "walked_rays_per_step": main_run["walked"] / args.steps,
            "scanned_rays_per_step": main_run["scanned"] / args.steps,
            "traversal": ("certified walk over the wide tree (bounce and shadow rays), a pinhole camera's rays from the "
                          "triangle lists of their cells on the image plane, order-replaying traversal for the few they leave over"
                          if main_run["walked"] else
                          "every ray tests every primitive of this small scene, without divisions, and its answer is certified "
                          "against the reference's leaf boxes (bounce and shadow rays); a pinhole camera's rays from the triangle "
                          "lists of their cells on the image plane; order-replaying traversal for the few they leave over"
                          if main_run.get("scanned") else
                          "order-replaying traversal (no wide tree for this scene, or --exact-only); a pinhole camera's rays "
                          "from the triangle lists of their cells on the image plane unless --exact-only"),
            "samples_per_s": m["samples_per_s"],
            "ms_per_step": m["ms_per_step"],
            "cold_ms_first_step": cold_ms,
            "data": "synthetic camera rays over the reference's shipped scene file (scenes/%s), per-(pixel,block) "
                    "seeded streams" % w["scene"],
            "config": {"workload": label,
                       "samples_per_stream": args.samples_per_stream,
                       "rays_per_sample": main_run["queries"] / main_run["samples"]},
        })
        if world > 1:
            line["render_ms_per_rank"] = m["render_ms_per_rank"]
            line["weak" if args.scaling == "strong" else "strong"] = summary(other_run)
            line["gather_ms"] = gather_ms
            line["gather_bytes_per_rank"] = gatherer.bytes_per_rank
        if excl is not None:
            line["roofline"] = roofline_render(w, excl, count, m)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_render(w, host_threads())
    ctx.close()
    return line


def roofline_render(w, excl, count, m):
    """The dominant kernel against the HBM roofline the metric prescribes, with
    the resources that actually bind (L2-served bytes, vector-ALU issue) beside
    it.  `excl`: stats of a render on ONE stream with every launch timed, so a
    launch's duration is its own; `count`: stats of a render that counted node
    visits and primitive tests on the device."""
    # The traversal kernel's own rays: what is traced at all, less the camera rays that k_camera_cells answers
    # from the lists of their cells (its launches are timed apart, as are those of the few rays left over).
    all_traced = excl["rays"] - excl["early_rays"]
    traced = all_traced - excl["listed_rays"]
    launches = max(excl["intersect_launches"], 1)
    avg_ms = excl["intersect_ms"] / launches
    # algorithmic bytes of the queries that walked the tree, from the device's own counts of
    # what they visited (closest-hit and occlusion traversals share the counters)
    queries = count["rays"] - count["early_rays"] - count["listed_rays"] + count["shadow_rays"]
    alg_bytes = (64.0 * count["inner_visits"] + 36.0 * (count["tri_tests"] - count["cell_tests"]) + 16.0 * count["sph_tests"] +
                 40.0 * queries)
    closest_share = (count["rays"] - count["early_rays"] - count["listed_rays"]) / max(queries, 1)
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
           "kernel": w.get("kernel", "k_intersect"), "avg_launch_ms": avg_ms, "launches": launches,
           "rays_per_launch": rays_per_launch,
           "rays_of_the_kernel": "the closest-hit queries that are traced at all less the camera rays answered from their "
                                 "cells' triangle lists by k_camera_cells (rta_stats.listed_rays), whose launches and those of "
                                 "k_intersect_odd are timed apart (exclusive_pass)",
           "bytes_per_ray": w["survey_bytes_per_ray"],
           "bytes_per_ray_source": "64 B x inner visits + 36 B x triangle tests + 40 B with the reference traversal's "
                                   "visit counts averaged over every closest-hit query of the workload",
           "timing": "HIP events around every launch of an extra render on ONE stream (nothing overlaps), untimed for "
                     "`value`",
           "frac_note": ("`achieved` prices every query at the reference traversal's algorithmic bytes; the kernel serves "
                         "them from its caches (`traffic`, `hbm_frac_from_counters`, `l2_frac`), which is how the figure "
                         "can pass the HBM peak on a scene of a few kilobytes"
                         + ("; this scene's few dozen primitives are not traversed at all but tested one and all from scalar "
                            "operands, and the kernel is bound by vector-ALU issue (about 75 instructions per triangle and wave, "
                            "every lane in use)" if w.get("kernel") == "k_intersect_flat" else "")) if achieved > HBM_PEAK_GBS else None,
           "exclusive_pass": {"gpu_ms": excl["gpu_ms"], "intersect_ms": excl["intersect_ms"],
                              "camera_cells_ms": excl["camera_ms"], "camera_cells_launches": excl["camera_launches"],
                              "listed_camera_rays": excl["listed_rays"], "leftover_ms": excl["leftover_ms"],
                              "advance_ms": excl["advance_ms"], "shadow_ms": excl["shadow_ms"],
                              "advance_launches": excl["advance_launches"],
                              "traced_mrays_per_s_of_the_kernel_alone": traced / max(excl["intersect_ms"], 1e-9) / 1e3,
                              "traced_mrays_per_s_of_the_closest_hit_launches": all_traced / max(
                                  excl["intersect_ms"] + excl["camera_ms"] + excl["leftover_ms"], 1e-9) / 1e3,
                              "kernel_time_over_gpu_time": (excl["intersect_ms"] + excl["camera_ms"] + excl["leftover_ms"] +
                                                            excl["advance_ms"] + excl["shadow_ms"]) / max(excl["gpu_ms"], 1e-9)},
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
            k = pj["kernels"].get(w.get("kernel", "k_intersect"), {})
            if "hbm_bytes_per_launch" in k:
                # the kernel's HBM bytes per render of the profiled run (same workload; its launches may be cut
                # differently: four slot groups there, one here), over this run's launches
                out["traffic"] = k["hbm_bytes_per_launch"] * k["launches"] / max(pj.get("renders_per_profiled_run", 1), 1) / launches
                out["traffic_over_algorithmic"] = out["traffic"] / per_launch if per_launch else None
                out["hbm_frac_from_counters"] = out["traffic"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_ms > 0 else None
            out["counters"] = {"source": "profiles/pmc_roofline.json (%s)" % pj.get("tag"), "kernels": pj["kernels"]}
    return out


# ------------------------------------------------------------------ ray batches

def bench_soup(args, env):
    """The intersect-only configuration.  Every rank holds the whole scene (0.7 GB of tree and
    triangles against 288 GB of HBM) and traces its equal slice of each batch; nothing is exchanged
    but the counts."""
    rank, world, local, dev, dist = env["rank"], env["world"], env["local"], env["dev"], env["dist"]
    from raytracer_amd import synthetic
    w = dict(WORKLOADS["soup10m"])
    n_tris = args.tris or w["tris"]
    n_rays = args.rays or w["rays"]
    side = int(round(n_rays ** 0.5))

    soup = synthetic.make_soup(n_tris)
    ctx = rta.Context(local)
    scene = rta.Scene.from_triangles(soup)
    t0 = time.perf_counter()
    build_kernels_ms = scene.build_bvh_on(ctx)
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx.upload(scene)
    upload_s = time.perf_counter() - t0
    d = scene.desc()
    n_nodes = int(d.n_nodes)

    def mine(rays):      # this rank's equal slice of a batch
        n = rays.shape[0]
        a, b = n * rank // world, n * (rank + 1) // world
        return rays[a:b].contiguous() if world > 1 else rays

    batches = {"coherent": mine(synthetic.coherent_rays(side, dev)), "incoherent": mine(synthetic.incoherent_rays(side * side, dev))}
    hits = {k: torch.empty((v.shape[0], 4), dtype=torch.float32, device=dev) for k, v in batches.items()}
    torch.cuda.synchronize()
    flags = dict(exact_only=args.exact_only)

    def step():
        out = {}
        for name, rays in batches.items():
            ctx.trace_closest_device(rays.data_ptr(), rays.shape[0], hits[name].data_ptr(), **flags)
            out[name] = ctx.last_batch_stats()
        return out

    for _ in range(args.warmup):
        step()
    fence(dist)
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence(dist)
    elapsed = time.perf_counter() - t0
    mine_rays = float(sum(v.shape[0] for v in batches.values())) * args.steps
    mx, sm, _ = over_ranks([elapsed, mine_rays], dev, dist)
    elapsed, total_rays = mx[0], sm[1]

    def stage_ms(name, key):
        return sum(st[name][key] for st in stats) / args.steps

    per_batch = {}
    for name, rays in batches.items():
        n = rays.shape[0]
        call_ms = stage_ms(name, "order_ms") + stage_ms(name, "walk_ms") + stage_ms(name, "replay_ms")
        per_batch[name] = {"rays": n, "order_ms": stage_ms(name, "order_ms"), "walk_ms": stage_ms(name, "walk_ms"),
                           "replay_ms": stage_ms(name, "replay_ms"), "mrays_per_s": n / call_ms / 1e3 if call_ms > 0 else None,
                           "sorted": bool(stats[-1][name]["sorted"]), "walked": bool(stats[-1][name]["walked"]),
                           "retraced_rays": stats[-1][name]["retraced_rays"],
                           "hit_fraction": float((hits[name][:, 0].view(torch.int32) >= 0).float().mean())}

    line = None
    if rank == 0:
        line = base_line(args, env, w)
        line.update({
            "value": total_rays / elapsed / 1e6,
            "value_counts": "closest-hit queries (rta_trace_closest: one per NearestHit call), both batches, all ranks; the "
                            "time includes the sort of the tracing order and the exact re-trace of the rays the walk hands on",
            "ms_per_step": elapsed / args.steps * 1e3,
            "scaling": "strong",
            "batches": per_batch,
            "scene": {"triangles": n_tris, "nodes": n_nodes, "bvh_build_kernels_ms": build_kernels_ms, "bvh_build_call_s": build_s,
                      "upload_s": upload_s},
            "data": "synthetic: seed-fixed random triangle soup (centres uniform in [-1,1]^3, edges within 0.01), a %dx%d pinhole "
                    "grid from (-3,0,0) and %d rays with uniform origins and directions" % (side, side, side * side),
            "config": {"workload": w["label"] if (n_tris, n_rays) == (w["tris"], w["rays"]) else
                       "%d-triangle random soup, %d coherent + %d incoherent rays, closest hit" % (n_tris, side * side, side * side),
                       **({"split": "every batch cut into %d equal slices, one per GPU, scene replicated" % world} if world > 1 else {})},
        })
        if not args.no_roofline:
            line["roofline"] = roofline_soup(ctx, batches, hits, per_batch, n_tris, n_nodes)
        if world == 1 and not args.no_cpu_baseline:
            n_check = WORKLOADS["soup10m"]["cpu_rays"]
            subsets = {}
            for name, rays in batches.items():
                stride = max(1, rays.shape[0] // n_check)
                subsets[name] = (rays[::stride].cpu().numpy(), hits[name][::stride].cpu().numpy(), stride)
            line["cpu_baseline"] = cpu_baseline_soup(soup, subsets, n_check, host_threads())
    ctx.close()
    return line


def roofline_soup(ctx, batches, hits, per_batch, n_tris, n_nodes):
    """The walk's kernel on the incoherent batch (the longest launch of a step) against the HBM
    roofline.  Algorithmic bytes per ray as the metric defines them - 64 B per inner-node visit and
    36 B per triangle test of the REFERENCE's traversal, + 40 B - with that traversal's visits counted
    on the device over the whole batch (the order-replaying kernel makes exactly the reference's
    visits); beside it what the walk itself reads, from its own counts."""
    out = {}
    for name, rays in batches.items():
        n = rays.shape[0]
        ctx.trace_closest_device(rays.data_ptr(), n, hits[name].data_ptr(), count_work=True, exact_only=True)
        ref = ctx.last_batch_stats()
        ctx.trace_closest_device(rays.data_ptr(), n, hits[name].data_ptr(), count_work=True)
        wk = ctx.last_batch_stats()
        ref_bytes = (64.0 * ref["replay_node_visits"] + 36.0 * ref["replay_prim_tests"]) / n + 40.0
        # a wide node is one 64-byte line, a triangle record 48 bytes, the leaf box that certifies a hit 32,
        # the ray 24 + 4 of its place in the tracing order, the hit 16
        walk_bytes = (64.0 * wk["walk_node_visits"] + 48.0 * wk["walk_tri_tests"] +
                      64.0 * wk["replay_node_visits"] + 48.0 * wk["replay_prim_tests"]) / n + 32.0 * per_batch[name]["hit_fraction"] + 44.0
        ms = per_batch[name]["walk_ms"]
        out[name] = {"reference_inner_visits_per_ray": ref["replay_node_visits"] / n,
                     "reference_tri_tests_per_ray": ref["replay_prim_tests"] / n,
                     "bytes_per_ray": ref_bytes,
                     "walk_node_visits_per_ray": wk["walk_node_visits"] / n, "walk_tri_tests_per_ray": wk["walk_tri_tests"] / n,
                     "walk_bytes_per_ray": walk_bytes,
                     "walk_ms": ms, "achieved_gbs": ref_bytes * n / (ms * 1e-3) / 1e9 if ms > 0 else None,
                     "walk_read_gbs": walk_bytes * n / (ms * 1e-3) / 1e9 if ms > 0 else None}
    dom = out["incoherent"]
    n = batches["incoherent"].shape[0]
    res = {"bound": "hbm", "achieved": dom["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": dom["achieved_gbs"] / HBM_PEAK_GBS if dom["achieved_gbs"] else None, "traffic": None,
           "kernel": "k_batch_walk (incoherent batch)", "avg_launch_ms": dom["walk_ms"], "rays_per_launch": n,
           "bytes_per_ray": dom["bytes_per_ray"],
           "bytes_per_ray_source": "64 B x inner visits + 36 B x triangle tests of the reference's traversal (counted on the "
                                   "device over the whole batch by the order-replaying kernel) + 40 B ray in and hit out",
           "timing": "HIP events recorded by the library on the launch's own stream around the kernel, averaged over the timed steps",
           "walk_read_gbs": dom["walk_read_gbs"],
           "walk_read_frac_of_peak": dom["walk_read_gbs"] / HBM_PEAK_GBS if dom["walk_read_gbs"] else None,
           "walk_read_frac_of_achievable": dom["walk_read_gbs"] / HBM_ACHIEVABLE_GBS if dom["walk_read_gbs"] else None,
           "walk_read_source": "what the walk itself reads, from its own device counts: 64 B per wide node + 48 B per triangle "
                               "record + 32 B of leaf box per hit + 44 B ray, order index and hit; the tree (%.2f GB of nodes and "
                               "triangle records) is beyond L2 and the Infinity Cache" % ((n_nodes * 32 + n_tris * 64 + n_tris * 32) / 1e9),
           "batches": out}
    ppath = os.path.join(ROOT, "profiles", "pmc_roofline_soup.json")
    if os.path.exists(ppath):
        with open(ppath) as f:
            pj = json.load(f)
        k = pj.get("kernels", {}).get("k_batch_walk:incoherent")
        if k and pj.get("rays_per_batch") == n and "hbm_bytes_per_launch" in k:
            res["traffic"] = k["hbm_bytes_per_launch"]
            res["traffic_over_algorithmic"] = res["traffic"] / (dom["bytes_per_ray"] * n)
            res["hbm_frac_from_counters"] = res["traffic"] / (dom["walk_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if dom["walk_ms"] else None
            res["counters"] = {"source": "profiles/pmc_roofline_soup.json (%s)" % pj.get("tag"), "kernels": pj["kernels"]}
    return res


if __name__ == "__main__":
    main()
