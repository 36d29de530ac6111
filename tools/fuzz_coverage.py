import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import raytracer_amd as rta
from test_gpu_fuzz import _scene
gpu = rta.Context(0)
for seed in range(28):
    text, what = _scene(seed)
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    _, _, st = gpu.render_scene(psc, seed=seed, samples_per_stream=1 + seed % 4)
    print(seed, what, {k: st[k] for k in ("rays", "early_rays", "scanned_rays", "walked_rays", "listed_rays", "retraced_rays", "shadow_rays")})
