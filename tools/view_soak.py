#!/usr/bin/env python3
"""dragon.txt and cornell_box.txt from random views (position, aim, roll, lens angle, frame size), small frames, GPU against the
CPU checker: a soak for the per-cell triangle lists and what follows them.  tools/view_soak.py N [first seed]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import orc
import raytracer_amd as rta


def with_render_block(text, **kw):
    head, sep, tail = text.partition("}")
    for key, val in kw.items():
        pattern = rf"(\b{key}\s+)(\[[^\]]*\]|\S+)"
        if re.search(pattern, head):
            head = re.sub(pattern, lambda m: m.group(1) + val, head, count=1)
        else:
            head += f"\t{key} {val}\n"
    return head + sep + tail


def vec(v):
    return "[" + " ".join("%r" % float(x) for x in v) + "]"


gpu = rta.Context(0)
n, first = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
scenes = {"dragon.txt": (np.array([0.056, 0.0768, 1.172]), 1.644), "cornell_box.txt": (np.array([0.0, 0.0, 1.0]), 1.7)}
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    name = "dragon.txt" if seed % 3 else "cornell_box.txt"
    centre, radius = scenes[name]
    d = rng.standard_normal(3)
    d /= np.linalg.norm(d)
    inside = name == "cornell_box.txt" and rng.integers(0, 2)
    cam = centre + d * radius * (rng.uniform(0.1, 0.5) if inside else rng.uniform(1.05, 3.0))
    text = with_render_block(open(f"{orc.SCENES}/{name}").read(), camera=vec(cam), look_at=vec(centre + rng.uniform(-0.6, 0.6, 3)),
                             up=vec(rng.standard_normal(3)), fov="%r" % float(rng.uniform(8, 100)), width=str(int(rng.integers(40, 200))),
                             height=str(int(rng.integers(30, 120))), samples=str(int(rng.integers(2, 6))))
    osc = orc.Scene.parse(text).build_bvh()
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    sps = int(rng.integers(1, 4))
    want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=seed, samples_per_stream=sps)
    bgr, acc, st = gpu.render_scene(psc, seed=seed, samples_per_stream=sps)
    ok = st["rays"] == cn["rays"] and np.array_equal(np.ascontiguousarray(acc, np.float32).view(np.uint32), np.ascontiguousarray(want_acc, np.float32).view(np.uint32)) and np.array_equal(bgr, want_bgr)
    if not ok:
        bad += 1
        print("seed", seed, name, "FAILED", flush=True)
    if seed % 50 == 0:
        print("seed", seed, name, "listed", st["listed_rays"], "walked", st["walked_rays"], "scanned", st["scanned_rays"], "retraced", st["retraced_rays"], "failures so far", bad, flush=True)
print("views", first, "..", first + n - 1, "failures", bad)
sys.exit(1 if bad else 0)
