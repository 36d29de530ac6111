#!/usr/bin/env python3
"""One rank's share of the multi-GPU benchmark on a single GPU: shard 0 of N of
dragon.txt 1080p at 1024 x N samples per pixel (what `bench.py --gpus N` gives
every rank), to see that a rank's rate does not depend on N."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401

import raytracer_amd as rta

scene = rta.Scene.load(os.path.join(ROOT, "scenes", "dragon.txt")).build_bvh()
ctx = rta.Context(0)
ctx.upload(scene)
strong = os.environ.get("STRONG") == "1"      # the fixed 1024-spp frame split N ways instead
for n in (1, 2, 4, 8):
    scene.set_render(1920, 1080, 1024 if strong else 1024 * n)
    for rep in range(2):
        t0 = time.perf_counter()
        _, _, st = ctx.render_scene(scene, seed=69420, samples_per_stream=16, shard_index=0, shard_count=n)
        dt = time.perf_counter() - t0
    print(f"{'strong' if strong else 'weak'} shard 0 of {n}: {st['samples']} samples, {st['rays']} rays in {dt * 1e3:.1f} ms wall, {st['gpu_ms']:.1f} ms on "
          f"the GPU: {st['rays'] / st['gpu_ms'] / 1e6:.2f} Grays/s, {st['steps']} steps", flush=True)
