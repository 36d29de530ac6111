#!/usr/bin/env python3
"""Prices the traversal's design decisions on the hardware: runs the production
closest-hit kernel and the variants of raytracer_amd/csrc/variants.hip on the
same device-resident ray sets (dragon.txt, 1080p camera rays; bounce rays leaving
the first hits), checks that all return the same bits, and prints times.  Under
`rocprofv3 --pmc ...` the per-kernel counters say WHY they differ."""
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

NAMES = {0: "production (64 B pair records, LDS stack, reciprocal box test, one loop)",
         1: "stack in private memory",
         2: "division-based box test",
         3: "reference 32 B nodes + node-number stack",
         4: "nested loops (inner nodes to a leaf, then the whole leaf)",
         5: "struct-of-arrays nodes (14 four-byte gathers per visit)"}


def main():
    reps = int(os.environ.get("REPS", "5"))
    scene = rta.Scene.load(os.path.join(ROOT, "scenes", os.environ.get("SCENE", "dragon.txt"))).build_bvh()
    scene.set_render(1920, 1080, 1)
    ctx = rta.Context(0)
    ctx.upload(scene)
    cam = scene.camera()
    cam19 = np.concatenate([np.array(getattr(cam, k), np.float32) for k in
                            ("origin", "upper_left", "du", "dv", "lens_u", "lens_v")] + [np.float32([-1])])
    prim = raygen.camera_grid(cam19, 1920, 1080)
    order = np.arange(1920 * 1080).reshape(135, 8, 240, 8).transpose(0, 2, 1, 3).reshape(-1)
    prim = prim[order]                                   # 8x8 tiles, as the renderer walks them
    dev = torch.device("cuda", 0)
    first = ctx.trace_closest(prim)
    hit = first["prim"] >= 0
    rng = np.random.default_rng(0)
    p = prim[hit, 0:3] + prim[hit, 3:6] * first["t"][hit, None]
    d = rng.standard_normal((p.shape[0], 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    bounce = np.tile(np.concatenate([p, d], axis=1).astype(np.float32), (int(os.environ.get("COPIES", "8")), 1))
    sets = {"camera rays 1080p": prim, "bounce rays x8": bounce}
    if os.environ.get("OCTANT"):
        # the same bounce rays, runs of OCTANT consecutive ones ordered by direction octant
        run = int(os.environ["OCTANT"])
        octant = (bounce[:, 3] < 0).astype(np.int64) | ((bounce[:, 4] < 0).astype(np.int64) << 1) | ((bounce[:, 5] < 0).astype(np.int64) << 2)
        key = (np.arange(len(bounce)) // run) * 8 + octant
        sets[f"bounce x8, octants/{run}"] = bounce[np.argsort(key, kind="stable")]
    for label, rr in sets.items():
        rt = torch.from_numpy(np.ascontiguousarray(rr)).to(dev)
        n = rr.shape[0]
        base = None
        for variant in [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3,4,5").split(",")]:
            ht = torch.zeros((n, 4), dtype=torch.float32, device=dev)
            try:
                ctx.trace_variant_device(variant, rt.data_ptr(), n, ht.data_ptr())
            except rta.RtaError as e:
                print(f"{label:20s} variant {variant} refused: {e}", flush=True)
                continue
            best = 1e9
            for _ in range(reps):
                t0 = time.perf_counter()
                ctx.trace_variant_device(variant, rt.data_ptr(), n, ht.data_ptr())
                best = min(best, time.perf_counter() - t0)
            res = ht.cpu().numpy().view(np.uint32)
            if variant == 100:
                f = ht.cpu().numpy()
                trips = f[:, 0].view(np.int32).astype(np.float64)
                if os.environ.get("SAVE_UNITS"):
                    np.save(os.path.join(ROOT, "gpurun_out", "units_" + label.split()[0] + ".npy"),
                            f[:len(f) // (8 if label.startswith("bounce") else 1), 1].astype(np.uint16))
                print(f"{label:20s} trips per wave {trips[::64].mean():.1f} (max {trips.max():.0f}), units per ray {f[:, 1].mean():.1f}, "
                      f"trips with both branches {f[::64, 2].mean():.1f}, lane occupancy {f[:, 1].sum() / trips.sum():.3f}", flush=True)
                continue
            if base is None:
                base = res
            same = np.array_equal(res, base)
            print(f"{label:20s} variant {variant} {NAMES.get(variant, 'experiment'):62s} {best * 1e3:8.3f} ms {n / best / 1e9:6.2f} Grays/s"
                  f"  same_bits={same}", flush=True)
            assert same


if __name__ == "__main__":
    main()
