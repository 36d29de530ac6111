#!/usr/bin/env python3
"""Times rta_trace_closest on ray sets of different coherence (device-resident
rays, HIP-synchronous calls).  A tuning probe, not a test."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import rays as raygen
import raytracer_amd as rta


def timed(ctx, rays_t, hits_t, reps=5):
    n = rays_t.shape[0]
    ctx.trace_closest_device(rays_t.data_ptr(), n, hits_t.data_ptr())
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.trace_closest_device(rays_t.data_ptr(), n, hits_t.data_ptr())
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    scene = rta.Scene.load(os.path.join(ROOT, "scenes", "dragon.txt")).build_bvh()
    scene.set_render(1920, 1080, 1)
    ctx = rta.Context(0)
    ctx.upload(scene)
    cam = scene.camera()
    cam19 = np.concatenate([np.array(getattr(cam, k), np.float32) for k in
                            ("origin", "upper_left", "du", "dv", "lens_u", "lens_v")] + [np.float32([-1])])
    prim = raygen.camera_grid(cam19, 1920, 1080)
    dev = torch.device("cuda", 0)

    def run(label, rr):
        rt = torch.from_numpy(np.ascontiguousarray(rr)).to(dev)
        ht = torch.empty((rr.shape[0], 4), dtype=torch.float32, device=dev)
        dt = timed(ctx, rt, ht)
        hit = (ht[:, 0].view(torch.int32) >= 0).float().mean().item()
        print(f"{label:34s} n={rr.shape[0]:9d} {dt * 1e3:8.3f} ms {rr.shape[0] / dt / 1e9:7.3f} Grays/s  hit={hit:.3f}", flush=True)
        return ht.cpu().numpy()

    h = run("primary row-major", prim)
    hit = h[:, 0].view(np.int32) >= 0
    # 8x8 tile order
    idx = np.arange(1920 * 1080).reshape(1080, 1920)
    tiled = idx.reshape(135, 8, 240, 8).transpose(0, 2, 1, 3).reshape(-1)
    run("primary 8x8 tiles", prim[tiled])
    run("primary hits only (row-major)", prim[hit])
    run("primary hits only x4", np.tile(prim[hit], (4, 1)))
    run("primary misses only", prim[~hit])
    rng = np.random.default_rng(0)
    run("primary shuffled", prim[rng.permutation(prim.shape[0])])
    # secondary: from hit points, random hemisphere-ish directions
    t = h[hit, 1:2]
    p = prim[hit, 0:3] + prim[hit, 3:6] * t
    d = rng.standard_normal((p.shape[0], 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sec = np.concatenate([p, d], axis=1).astype(np.float32)
    run("secondary random dirs", sec)
    run("secondary random dirs x4", np.tile(sec, (4, 1)))
    lo, hi = np.float32([-0.6, -1.3, 0.2]), np.float32([0.7, 1.4, 2.1])
    run("incoherent in-box", raygen.random_rays(2_000_000, lo, hi, 5))


if __name__ == "__main__":
    main()
