#!/usr/bin/env python3
"""Renders dragon.txt with the camera turned away from the model (every sample
ends in the sky): what the advance kernel costs per sample on its shortest path.
Run under `rocprofv3 --pmc SQ_INSTS_VALU ...` for instruction counts."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (its HIP runtime must load first)

import raytracer_amd as rta

text = open(os.path.join(ROOT, "scenes", "dragon.txt")).read()
scene = rta.Scene.parse(text).build_bvh()
scene.set_render(1920, 1080, int(os.environ.get("SPP", "128")))
s = scene.settings
if os.environ.get("AWAY", "1") == "1":
    for k in range(3):
        s.look_at[k] = 2 * s.camera[k] - s.look_at[k]
    scene.settings = s
ctx = rta.Context(0)
ctx.upload(scene)
for _ in range(2):
    t0 = time.perf_counter()
    _, _, st = ctx.render_scene(scene, seed=1, samples_per_stream=32)
    dt = time.perf_counter() - t0
print(f"samples {st['samples']} rays {st['rays']} in {dt * 1e3:.1f} ms: {st['samples'] / dt / 1e9:.2f} Gsamples/s", flush=True)
