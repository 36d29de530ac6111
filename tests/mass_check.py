#!/usr/bin/env python3
"""Bulk parity run (not collected by pytest): many random and camera rays through
the GPU's closest-hit and occlusion entry points against the CPU checker, bit
for bit.  Rare events (box decisions that need the exact quotients, ties) only
show up in bulk.

    python tests/mass_check.py [--rays 64] (millions per scene and kind)
"""
import argparse
import os
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np
import torch  # noqa: F401

import orc
import rays as raygen
import raytracer_amd as rta


def check(name, millions, threads):
    path = os.path.join(orc.SCENES, name)
    osc = orc.Scene.load(path).build_bvh()
    psc = rta.Scene.load(path).build_bvh()
    ctx = rta.Context(0)
    ctx.upload(psc)
    nodes = osc.nodes()
    lo, hi = nodes["lo"][0], nodes["hi"][0]
    pad = (hi - lo) * 0.25
    chunk = 1 << 20
    bad = 0
    total = 0
    with ThreadPoolExecutor(threads) as pool:
        for part in range(millions):
            rng = np.random.default_rng(1000 + part)
            rr = raygen.random_rays(chunk, lo - pad, hi + pad, seed=5000 + part).astype(np.float32)
            if part % 4 == 1:
                # rays that start ON the geometry: bounce-like, many zero entry distances
                got0 = ctx.trace_closest(rr)
                hit = got0["prim"] >= 0
                rr[hit, 0:3] = rr[hit, 0:3] + rr[hit, 3:6] * got0["t"][hit, None]
                d = rng.standard_normal((chunk, 3)).astype(np.float32)
                rr[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
            if part % 4 == 2:
                # axis-parallel and grid-aligned rays: ties between boxes
                rr[:, 3:6] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, chunk)] * rng.choice([-1, 1], (chunk, 1))
                rr[:, 0:3] = np.round(rr[:, 0:3] * 8) / 8
            dist = (rng.random(chunk, dtype=np.float32) * np.float32(np.linalg.norm(hi - lo))).astype(np.float32)
            got = ctx.trace_closest(rr)
            occ = ctx.trace_occluded(rr, dist)
            pieces = np.array_split(np.arange(chunk), threads)
            want = list(pool.map(lambda ix: osc.trace_closest(rr[ix]), pieces))
            wocc = list(pool.map(lambda ix: osc.trace_occluded(rr[ix], dist[ix]), pieces))
            idx = np.concatenate([w[0] for w in want])
            t = np.concatenate([w[1] for w in want])
            u = np.concatenate([w[2] for w in want])
            v = np.concatenate([w[3] for w in want])
            same = (got["prim"] == idx) & (got["t"].view(np.uint32) == t.view(np.uint32))
            tri = idx >= 0
            same &= ~tri | ((got["u"].view(np.uint32) == u.view(np.uint32)) & (got["v"].view(np.uint32) == v.view(np.uint32)))
            same &= np.asarray(occ, bool) == np.concatenate(wocc).astype(bool)
            bad += int((~same).sum())
            total += chunk
            if not same.all():
                k = int(np.nonzero(~same)[0][0])
                print(f"MISMATCH {name} part {part} ray {rr[k].tolist()} gpu {got['prim'][k]} {got['t'][k]} cpu {idx[k]} {t[k]}",
                      flush=True)
            if part % 8 == 7 or part == millions - 1:
                print(f"{name}: {total} rays, {bad} mismatches", flush=True)
    ctx.close()
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=64)
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    args = ap.parse_args()
    bad = 0
    for name in ("dragon.txt", "cornell_box.txt"):
        bad += check(name, args.rays, args.threads)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
