#!/usr/bin/env python3
"""Work counts of the certified walk on the triangle soup: tools/probe_soup.py [--tris N] [--rays N]
(RTA_WIDE_COUNT=1 makes rta_debug_trace_variant 20 print node visits and triangle tests per ray)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import raytracer_amd as rta
from soup_bench import coherent_rays, incoherent_rays, make_soup

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=10_000_000)
ap.add_argument("--rays", type=int, default=16 * 1024 * 1024)
args = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = rta.Context(0)
scene = rta.Scene.from_triangles(make_soup(args.tris))
scene.build_bvh_on(ctx)
ctx.upload(scene)
side = int(round(args.rays ** 0.5))
for name, rays in (("coherent", coherent_rays(side, dev)), ("incoherent", incoherent_rays(args.rays, dev))):
    n = rays.shape[0]
    hits = torch.empty((n, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for variant in (0, 20):
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            ctx.trace_variant_device(variant, rays.data_ptr(), n, hits.data_ptr())
            best = min(best, time.perf_counter() - t0)
        print(f"{name} variant {variant}: {n / best / 1e6:.0f} Mrays/s", flush=True)
