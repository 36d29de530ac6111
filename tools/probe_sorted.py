#!/usr/bin/env python3
"""What is tracing bounce rays in a sorted order worth on a tree that fits the
caches?  Builds first- and second-bounce rays of dragon.txt (cosine-distributed
about the geometric normal of the hit triangle), and times rta_trace_closest on
them in the order the renderer would hold them (by pixel), shuffled, and sorted
by several keys.  A tuning probe, not a test."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import rays as raygen
import raytracer_amd as rta


def spread(v, bits):
    out = np.zeros_like(v)
    for b in range(bits):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def keys(rr, lo, hi, bits, octant_major=False):
    cell = np.clip((rr[:, 0:3] - lo) / (hi - lo) * (1 << bits), 0, (1 << bits) - 1).astype(np.int64)
    morton = spread(cell[:, 0], bits) | (spread(cell[:, 1], bits) << 1) | (spread(cell[:, 2], bits) << 2)
    octant = (rr[:, 3] < 0).astype(np.int64) | ((rr[:, 4] < 0).astype(np.int64) << 1) | ((rr[:, 5] < 0).astype(np.int64) << 2)
    return (octant << (3 * bits)) | morton if octant_major else (morton << 3) | octant


def bounce_from(rr, hits, geom, rng):
    hit = hits["prim"] >= 0
    rr, hits = rr[hit], hits[hit]
    p = rr[:, 0:3] + rr[:, 3:6] * hits["t"][:, None]
    g = geom[hits["prim"]]
    n = np.cross(g[:, 0:3] - g[:, 6:9], g[:, 3:6] - g[:, 6:9])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n[np.einsum("ij,ij->i", n, rr[:, 3:6]) > 0] *= -1
    # cosine-weighted about n
    a = np.where(np.abs(n[:, 0:1]) > 0.9, np.float32([[0, 1, 0]]), np.float32([[1, 0, 0]]))
    y = np.cross(n, a)
    y /= np.linalg.norm(y, axis=1, keepdims=True)
    x = np.cross(n, y)
    r1, r2 = rng.random(len(p)), rng.random(len(p))
    phi = 2 * np.pi * r1
    l = np.stack([np.cos(phi) * np.sqrt(r2), np.sin(phi) * np.sqrt(r2), np.sqrt(1 - r2)], axis=1)
    d = x * l[:, 0:1] + y * l[:, 1:2] + n * l[:, 2:3]
    return np.concatenate([p, d], axis=1).astype(np.float32)


def main():
    reps = int(os.environ.get("REPS", "5"))
    scene = rta.Scene.load(os.path.join(ROOT, "scenes", "dragon.txt")).build_bvh()
    scene.set_render(1920, 1080, 1)
    geom = scene.arrays()["prim_geom"]
    nodes = scene.arrays()["node_bounds"]
    lo, hi = nodes[0, 0:3], nodes[0, 3:6]
    ctx = rta.Context(0)
    ctx.upload(scene)
    cam = scene.camera()
    cam19 = np.concatenate([np.array(getattr(cam, k), np.float32) for k in
                            ("origin", "upper_left", "du", "dv", "lens_u", "lens_v")] + [np.float32([-1])])
    prim = raygen.camera_grid(cam19, 1920, 1080)
    order = np.arange(1920 * 1080).reshape(135, 8, 240, 8).transpose(0, 2, 1, 3).reshape(-1)
    prim = prim[order]
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    first = ctx.trace_closest(prim)
    b1 = np.concatenate([bounce_from(prim, first, geom, rng) for _ in range(8)])
    h1 = ctx.trace_closest(b1)
    b2 = bounce_from(b1, h1, geom, rng)
    b2 = np.concatenate([b2] * 1)

    def run(label, rr):
        rt = torch.from_numpy(np.ascontiguousarray(rr)).to(dev)
        n = rr.shape[0]
        ht = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        ctx.trace_closest_device(rt.data_ptr(), n, ht.data_ptr())
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.trace_closest_device(rt.data_ptr(), n, ht.data_ptr())
            best = min(best, time.perf_counter() - t0)
        print(f"{label:44s} n={n:8d} {best * 1e3:8.3f} ms {n / best / 1e9:6.2f} Grays/s", flush=True)

    for name, rr in (("bounce 1", b1), ("bounce 2", b2)):
        run(f"{name} by pixel", rr)
        run(f"{name} shuffled", rr[rng.permutation(len(rr))])
        for bits in (4, 5, 7, 10):
            run(f"{name} sorted morton{bits}|octant", rr[np.argsort(keys(rr, lo, hi, bits), kind="stable")])
        run(f"{name} sorted octant|morton7", rr[np.argsort(keys(rr, lo, hi, 7, True), kind="stable")])
        run(f"{name} sorted octant|morton10", rr[np.argsort(keys(rr, lo, hi, 10, True), kind="stable")])
        # sorted within runs of 64 Ki rays only (what a per-launch partial sort could do)
        k = keys(rr, lo, hi, 7)
        run(f"{name} sorted in runs of 65536", rr[np.argsort((np.arange(len(rr)) // 65536 << 24) | k, kind="stable")])


if __name__ == "__main__":
    main()
