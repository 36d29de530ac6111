#!/usr/bin/env python3
"""dragon.txt standing on a floor of two large triangles: does the scene get a wide tree, what does a frame cost
with and without it, and are the bits the checker's?   tools/probe_floor.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401

import orc
import raytracer_amd as rta

text = open(os.path.join(ROOT, "scenes", "dragon.txt")).read()
floor = "\nmaterial {\n\tcolor [0.6 0.6 0.6]\n\troughness 1\n}\n\ntriangle_mesh {\n\tvertices [ -4 -4 0.27  4 -4 0.27  4 4 0.27   -4 -4 0.27  4 4 0.27  -4 4 0.27 ]\n}\n"
text += floor
ctx = rta.Context(0)
for w, h, spp, check in ((96, 54, 16, True), (1920, 1080, 256, False)):
    t = text.replace("width 400", f"width {w}").replace("height 400", f"height {h}").replace("samples 25", f"samples {spp}")
    psc = rta.Scene.parse(t).build_bvh()
    ctx.upload(psc)
    print("wide tree:", bool(ctx.scene_buffer(6).view(np.uint32)[11]), flush=True)
    for flags in (0, rta.RENDER_EXACT_ONLY):
        for _ in range(2):
            t0 = time.perf_counter()
            bgr, acc, st = ctx.render_scene(psc, seed=69420, samples_per_stream=8, flags=flags)
            dt = time.perf_counter() - t0
        print(f"{w}x{h}x{spp} flags {flags}: {st['gpu_ms']:.1f} ms, rays {st['rays']}, walked {st['walked_rays']}, retraced {st['retraced_rays']}, "
              f"early {st['early_rays']}", flush=True)
    if check:
        osc = orc.Scene.parse(t).build_bvh()
        want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=69420, samples_per_stream=8)
        print("bit-exact:", np.array_equal(acc.view(np.uint32), want_acc.view(np.uint32)), cn["rays"] == st["rays"])
