#!/usr/bin/env python3
"""Bulk parity run for whole frames (not collected by pytest): dragon.txt at
1920x1080 with many samples per pixel, GPU radiance sums against the CPU
checker's bit for bit.  Every sample that the GPU path answers on the image
plane (no ray) is computed in full by the checker, so a wrong early answer at
the silhouette shows as a differing pixel or ray count.

    python tests/frame_check.py [--spp 64] [--sps 16]
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np
import torch  # noqa: F401

import orc
import raytracer_amd as rta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="dragon.txt")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--sps", type=int, default=16)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    path = os.path.join(orc.SCENES, args.scene)
    osc = orc.Scene.load(path).build_bvh()
    psc = rta.Scene.load(path).build_bvh()
    for sc in (osc, psc):
        sc.set_render(1920, 1080, args.spp)
    ctx = rta.Context(0)
    ctx.upload(psc)
    bgr, acc, st = ctx.render_scene(psc, seed=4242, samples_per_stream=args.sps)
    print(f"GPU: {st['samples']} samples, {st['rays']} rays ({st['early_rays']} without a traversal), {st['gpu_ms']:.1f} ms",
          flush=True)
    t0 = time.perf_counter()
    want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=4242, samples_per_stream=args.sps,
                                        threads=args.threads)
    print(f"checker: {cn['samples']} samples, {cn['rays']} rays in {time.perf_counter() - t0:.1f} s", flush=True)
    differ = (acc.view(np.uint32) != want_acc.view(np.uint32)).any(axis=2)
    ok = not differ.any() and st["rays"] == cn["rays"] and np.array_equal(bgr, want_bgr)
    print(f"{args.scene} 1920x1080x{args.spp}: {int(differ.sum())} pixels differ, rays {'equal' if st['rays'] == cn['rays'] else 'DIFFER'}, "
          f"bytes {'equal' if np.array_equal(bgr, want_bgr) else 'DIFFER'}")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
