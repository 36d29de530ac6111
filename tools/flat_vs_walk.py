#!/usr/bin/env python3
"""Small scenes rendered with the current RTA_FLAT setting: gpu_ms of a 960x540 x 64 spp frame for scenes of 12 to 64
primitives (tests/test_gpu_flat.py's).  Run once per setting: RTA_FLAT=1 (scan only scenes without a wide tree) and 2."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import raytracer_amd as rta
from test_gpu_flat import _small_scene

gpu = rta.Context(0)
for n_tris, n_spheres, light in ((12, 0, False), (20, 3, True), (30, 2, True), (44, 4, True), (61, 0, True)):
    text = _small_scene(3, n_tris=n_tris, n_spheres=n_spheres, light=light).replace("width 48", "width 960").replace("height 40", "height 540").replace("samples 6", "samples 64")
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    best = 1e9
    for _ in range(3):
        _, _, st = gpu.render_scene(psc, seed=1, samples_per_stream=8, want_accum=False)
        best = min(best, st["gpu_ms"])
    print("RTA_FLAT=%s: %d triangles, %d spheres, light %d: %.2f ms, scanned %d walked %d retraced %d" %
          (os.environ.get("RTA_FLAT", "default"), n_tris + (2 if light else 0), n_spheres, light, best, st["scanned_rays"], st["walked_rays"], st["retraced_rays"]), flush=True)
