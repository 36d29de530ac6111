#!/usr/bin/env python3
"""Is the host's launch rate part of the render time?  The benchmark workload
with and without per-launch events."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401

import raytracer_amd as rta

scene = rta.Scene.load(os.path.join(ROOT, "scenes", "dragon.txt")).build_bvh()
scene.set_render(1920, 1080, 1024)
ctx = rta.Context(0)
ctx.upload(scene)
for flags in (0, rta.RENDER_TIME_KERNELS, 0, rta.RENDER_TIME_KERNELS):
    t0 = time.perf_counter()
    _, _, st = ctx.render_scene(scene, seed=69420, samples_per_stream=32, flags=flags)
    dt = time.perf_counter() - t0
    print(f"flags {flags}: {dt * 1e3:.1f} ms wall, {st['gpu_ms']:.1f} ms on the GPU, {st['steps']} steps, "
          f"{(st['rays']) / dt / 1e9:.2f} Grays/s", flush=True)
