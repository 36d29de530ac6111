#!/usr/bin/env python3
"""Shard 0 of 8 of dragon.txt 1080p x 1024 spp, three times (for kernel traces of a strong-scaled rank's share)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import raytracer_amd as rta
scene = rta.Scene.load(os.path.join(ROOT, "scenes", "dragon.txt")).build_bvh()
ctx = rta.Context(0)
ctx.upload(scene)
scene.set_render(1920, 1080, 1024)
n = int(os.environ.get("SHARDS", "8"))
for rep in range(3):
    _, _, st = ctx.render_scene(scene, seed=69420, samples_per_stream=int(os.environ.get("SPS", "16")), shard_index=0, shard_count=n,
                                streams=int(os.environ.get("STREAMS", "0")), slots=int(os.environ.get("SLOTS", "0")))
    print(f"{st['gpu_ms']:.2f} ms, {st['steps']} steps", flush=True)
