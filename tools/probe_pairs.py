#!/usr/bin/env python3
"""Times the closest-hit kernel variants of csrc/variants.hip on first- and
second-bounce rays of dragon.txt (see probe_sorted.py for how they are made)
and checks that every variant returns the production kernel's bits.
VARIANTS=0,6,7 REPS=5 python tools/probe_pairs.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import rays as raygen
import raytracer_amd as rta
from probe_sorted import bounce_from


def main():
    reps = int(os.environ.get("REPS", "5"))
    variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3,4,5").split(",")]
    scene = rta.Scene.load(os.path.join(ROOT, "scenes", "dragon.txt")).build_bvh()
    scene.set_render(1920, 1080, 1)
    geom = scene.arrays()["prim_geom"]
    ctx = rta.Context(0)
    ctx.upload(scene)
    cam = scene.camera()
    cam19 = np.concatenate([np.array(getattr(cam, k), np.float32) for k in
                            ("origin", "upper_left", "du", "dv", "lens_u", "lens_v")] + [np.float32([-1])])
    prim = raygen.camera_grid(cam19, 1920, 1080)
    order = np.arange(1920 * 1080).reshape(135, 8, 240, 8).transpose(0, 2, 1, 3).reshape(-1)
    prim = prim[order]
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    first = ctx.trace_closest(prim)
    b1 = np.concatenate([bounce_from(prim, first, geom, rng) for _ in range(8)])
    h1 = ctx.trace_closest(b1)
    b2 = np.concatenate([bounce_from(b1, h1, geom, rng) for _ in range(4)])
    big = int(os.environ.get("TILE", "1"))
    for label, rr in (("bounce 1", np.tile(b1, (big, 1))), ("bounce 2", np.tile(b2, (big, 1))), ("camera", prim)):
        rt = torch.from_numpy(np.ascontiguousarray(rr)).to(dev)
        n = rr.shape[0]
        base = None
        for variant in variants:
            ht = torch.zeros((n, 4), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            ctx.trace_variant_device(variant, rt.data_ptr(), n, ht.data_ptr())
            best = 1e9
            for _ in range(reps):
                t0 = time.perf_counter()
                ctx.trace_variant_device(variant, rt.data_ptr(), n, ht.data_ptr())
                best = min(best, time.perf_counter() - t0)
            res = ht.cpu().numpy().view(np.uint32)
            if base is None:
                base = res
            same = np.array_equal(res, base)
            print(f"{label:10s} variant {variant:3d} n={n:8d} {best * 1e3:8.3f} ms {n / best / 1e9:6.2f} Grays/s same_bits={same}", flush=True)


if __name__ == "__main__":
    main()
