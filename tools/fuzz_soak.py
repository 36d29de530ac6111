#!/usr/bin/env python3
"""tests/test_gpu_fuzz.py's two checks over a range of seeds (a soak, not collected by pytest): tools/fuzz_soak.py FIRST LAST"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import raytracer_amd as rta
import test_gpu_fuzz as fz

gpu = rta.Context(0)
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
    for check in (fz.test_a_random_scene_renders_as_the_checker_has_it, fz.test_ray_batches_over_a_random_scene):
        try:
            check.__wrapped__(gpu, seed) if hasattr(check, "__wrapped__") else check(gpu, seed)
        except AssertionError as e:
            bad += 1
            print("seed", seed, check.__name__, "FAILED:", str(e)[:300], flush=True)
    if seed % 100 == 0:
        print("seed", seed, "failures so far", bad, flush=True)
print("seeds", first, "..", last - 1, "failures", bad)
sys.exit(1 if bad else 0)
