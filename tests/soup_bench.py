#!/usr/bin/env python3
"""Intersect-only microbenchmark: a seed-fixed random triangle soup, coherent
and incoherent ray batches resident in HBM, closest-hit through the C ABI.

    python tests/soup_bench.py [--tris 10000000] [--rays 67108864] [--check 1048576]

(It lives under tests/ because it checks against the CPU checker; pytest does
not collect it.)

The tree is the reference builder's, built on the GPU
(rta_scene_build_bvh_device) and compared with the host builder's
(rta_scene_build_bvh; the reference binary itself stops at 16384 objects).  Correctness: a strided subset
of each batch is traced by the CPU checker and must agree bit for bit in
primitive index and t; a checksum over all indices is printed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import raytracer_amd as rta


from raytracer_amd.synthetic import coherent_rays, incoherent_rays, make_soup  # noqa: E402,F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--rays", type=int, default=64 * 1024 * 1024)
    ap.add_argument("--check", type=int, default=1 << 20, help="rays per batch verified against the CPU checker")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-host-build", action="store_true", help="skip the host builder's comparison tree")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)

    soup = make_soup(args.tris)
    ctx = rta.Context(0)
    # the reference builder's tree, built on the GPU ...
    scene = rta.Scene.from_triangles(soup)
    t0 = time.perf_counter()
    kernel_ms = scene.build_bvh_on(ctx)
    device_s = time.perf_counter() - t0
    d = scene.desc()
    print(f"soup: {d.n_prims} triangles, {d.n_nodes} nodes; tree built on the GPU in {kernel_ms:.1f} ms of kernels, "
          f"{device_s:.2f} s with transfers and the host-side reordering of the scene arrays", flush=True)
    build = {"device_kernels_ms": kernel_ms, "device_call_s": device_s}
    if not args.no_host_build:
        # ... and by the host builder, which must give the same arrays
        other = rta.Scene.from_triangles(soup)
        t0 = time.perf_counter()
        other.build_bvh()
        build["host_call_s"] = time.perf_counter() - t0
        a, b = scene.arrays(), other.arrays()
        build["same_tree"] = bool(all(np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k],
                                                     b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k])
                                      for k in a) and np.array_equal(scene.permutation(), other.permutation()))
        print(f"host builder: {build['host_call_s']:.2f} s, same tree: {build['same_tree']}", flush=True)
        del other, a, b
    t0 = time.perf_counter()
    ctx.upload(scene)
    build["upload_s"] = time.perf_counter() - t0
    print(f"validated, flattened and uploaded in {build['upload_s']:.2f} s", flush=True)

    checker = None
    if args.check:
        import orc
        t0 = time.perf_counter()
        checker = orc.Scene.from_triangles(soup).build_bvh()
        print(f"checker tree built in {time.perf_counter() - t0:.1f} s, depth {checker.depth()}", flush=True)

    side = int(round(args.rays ** 0.5))
    batches = {"coherent": coherent_rays(side, dev), "incoherent": incoherent_rays(args.rays, dev)}
    report = {"triangles": int(d.n_prims), "nodes": int(d.n_nodes), "build": build, "batches": {}}
    for name, rays in batches.items():
        n = rays.shape[0]
        hits = torch.empty((n, 4), dtype=torch.float32, device=dev)
        ctx.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
        best = 1e9
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
            best = min(best, time.perf_counter() - t0)
        prim = hits[:, 0].view(torch.int32)
        entry = {"rays": n, "ms": best * 1e3, "mrays_per_s": n / best / 1e6,
                 "hit_fraction": float((prim >= 0).float().mean()),
                 "index_checksum": int((prim.long() * (torch.arange(n, device=dev) % 1000003 + 1)).sum().item())}
        if checker is not None:
            stride = max(1, n // args.check)
            sub = rays[::stride].cpu().numpy()
            idx, t, _, _, cn = checker.trace_closest(sub, counters=True)
            got = hits[::stride].cpu().numpy()
            same = np.array_equal(got[:, 0].view(np.int32), idx) and np.array_equal(got[:, 1].view(np.uint32),
                                                                                     t.view(np.uint32))
            entry.update({"checked": int(len(sub)), "bit_exact": bool(same),
                          "inner_visits_per_ray": float(cn[3]) / len(sub), "tri_tests_per_ray": float(cn[4]) / len(sub)})
            entry["bytes_per_ray"] = 64 * entry["inner_visits_per_ray"] + 36 * entry["tri_tests_per_ray"] + 40
            entry["algorithmic_gb_per_s"] = entry["bytes_per_ray"] * n / best / 1e9
            entry["hbm_fraction_of_8tbs"] = entry["algorithmic_gb_per_s"] / 8000.0
            if not same:
                bad = np.nonzero(got[:, 0].view(np.int32) != idx)[0]
                print(f"MISMATCH in {name}: {len(bad)} indices differ, first ray {sub[bad[0]] if len(bad) else None}")
        report["batches"][name] = entry
        print(name, json.dumps(entry), flush=True)
        del hits
    print(json.dumps(report))
    ok = all(b.get("bit_exact", True) for b in report["batches"].values()) and build.get("same_tree", True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
