#!/usr/bin/env python3
"""Work counts of a dragon.txt 1080p render (RTA_RENDER_COUNT_WORK): node visits, triangle tests and, in `sph_tests`
on this sphere-less scene, whatever census the library build under test adds there."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import raytracer_amd as rta
scene = rta.Scene.load(os.path.join(ROOT, "scenes", os.environ.get("SCENE", "dragon.txt"))).build_bvh()
ctx = rta.Context(0)
ctx.upload(scene)
scene.set_render(1920, 1080, int(os.environ.get("SPP", "64")))
for flags in (rta.RENDER_COUNT_WORK, 0):
    _, _, st = ctx.render_scene(scene, seed=69420, samples_per_stream=8, flags=flags, want_accum=False, want_bgr=False)
    traced = st["rays"] - st["early_rays"]
    print({k: st[k] for k in ("samples", "rays", "early_rays", "walked_rays", "listed_rays", "retraced_rays", "inner_visits", "tri_tests", "sph_tests", "cell_tests")},
          f"traced {traced}, per traced ray: {st['inner_visits'] / traced:.2f} visits, {st['tri_tests'] / traced:.2f} tests, {st['sph_tests'] / traced:.2f} census; {st['cell_tests'] / max(st['listed_rays'], 1):.2f} list tests per listed ray; {st['gpu_ms']:.1f} ms",
          flush=True)
