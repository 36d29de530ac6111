#!/usr/bin/env python3
"""Times rta_trace_occluded on dragon.txt: rays leaving the surface in random directions, limits drawn around
the distance of their nearest hit.  RTA_BATCH_EXACT=1 prices the replaying traversal on the same batch.
    REPS=5 TILE=8 python tools/probe_occluded.py"""
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

reps, tile = int(os.environ.get("REPS", "5")), int(os.environ.get("TILE", "8"))
scene = rta.Scene.load(os.path.join(ROOT, "scenes", "dragon.txt")).build_bvh()
scene.set_render(1920, 1080, 1)
geom = scene.arrays()["prim_geom"]
ctx = rta.Context(0)
ctx.upload(scene)
cam = scene.camera()
cam19 = np.concatenate([np.array(getattr(cam, k), np.float32) for k in
                        ("origin", "upper_left", "du", "dv", "lens_u", "lens_v")] + [np.float32([-1])])
prim = raygen.camera_grid(cam19, 1920, 1080)
rng = np.random.default_rng(0)
first = ctx.trace_closest(prim)
b1 = np.concatenate([bounce_from(prim, first, geom, rng) for _ in range(8)])
near = ctx.trace_closest(b1)["t"]
with np.errstate(over="ignore"):
    limit = np.where(near < 3e38, near * rng.choice(np.float32([0.5, 0.9, 1.1, 2.0]), near.shape[0]), np.float32(1.0)).astype(np.float32)
rr, limit = np.tile(b1, (tile, 1)), np.tile(limit, tile)
dev = torch.device("cuda", 0)
rt, lt = torch.from_numpy(np.ascontiguousarray(rr)).to(dev), torch.from_numpy(limit).to(dev)
out = torch.zeros(rr.shape[0], dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx.trace_occluded_device(rt.data_ptr(), lt.data_ptr(), rr.shape[0], out.data_ptr())
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter()
    ctx.trace_occluded_device(rt.data_ptr(), lt.data_ptr(), rr.shape[0], out.data_ptr())
    best = min(best, time.perf_counter() - t0)
print(f"occlusion queries n={rr.shape[0]} {best * 1e3:8.3f} ms {rr.shape[0] / best / 1e9:6.2f} G/s occluded {float(out.float().mean()):.3f} "
      f"checksum {int(out.long().sum().item())}", flush=True)
