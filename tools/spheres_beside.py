#!/usr/bin/env python3
"""dragon.txt with N small spheres scattered around it, 1920x1080 x 64 spp: the render with the walk (spheres tested one by one
for every ray) against the render through the order-replaying traversal alone.  tools/spheres_beside.py [N ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as rta

base = open(os.path.join(ROOT, "scenes", "dragon.txt")).read()
gpu = rta.Context(0)
for n in [int(x) for x in sys.argv[1:]] or [0, 8, 24, 48, 49]:
    rng = np.random.default_rng(1)
    balls = "".join("sphere {\n\tcenter [%r %r %r]\n\tradius %r\n}\n\n" % (*map(float, rng.uniform((-1, -1, 0.2), (1, 1, 2.2), 3)), float(rng.uniform(0.03, 0.08)))
                    for _ in range(n))
    psc = rta.Scene.parse(base + "\nmaterial {\n\tcolor [0.8 0.3 0.2]\n\troughness 0.3\n}\n\n" + balls).build_bvh()
    psc.set_render(1920, 1080, 64)
    gpu.upload(psc)
    out = []
    for flags in (0, rta.RENDER_EXACT_ONLY):
        best = 1e9
        for _ in range(2):
            _, _, st = gpu.render_scene(psc, seed=1, samples_per_stream=8, flags=flags, want_accum=False)
            best = min(best, st["gpu_ms"])
        out.append((best, st["walked_rays"], st["listed_rays"]))
    print("%d spheres: %.1f ms (walked %d, listed %d), replaying only %.1f ms" % (n, out[0][0], out[0][1], out[0][2], out[1][0]), flush=True)
