#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ by running the REAL reference.

Needs the reference built from its own sources (make -C oracle ref, which
compiles /root/reference/src into oracle/_ref/); run in the build container:

    python tests/golden/make_golden.py

Everything written here is data: inputs (rays, materials, seeds, malformed scene
snippets written by this script) and the reference's outputs for them.
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import orc  # noqa: E402
import rays as raygen  # noqa: E402

SCENES = ("cornell_box.txt", "dragon.txt")

BAD_SCENES = {
    "unknown_command": "foo { }\n",
    "missing_brace": "render width 4 }\n",
    "unknown_key": "render { widht 4 }\n",
    "bad_number": "sphere { radius 1.5x center [0 0 0] }\n",
    "bad_integer": "render { width 4.5 }\n",
    "integer_range": "render { width 70000 }\n",
    "negative_integer": "render { samples -3 }\n",
    "short_vector": "render { camera [1 2] }\n",
    "long_vector": "render { camera [1 2 3 4] }\n",
    "vector_not_array": "material { color 1 2 3 }\n",
    "vertices_not_array": "triangle_mesh { vertices 1 2 3 }\n",
    "truncated_triangle": "triangle_mesh { vertices [ 0 0 0 1 0 0 0 1 ] }\n",
    "eof_in_block": "render { width 4\n",
    "eof_in_array": "triangle_mesh { vertices [ 0 0 0 1 0 0 0 1 0\n",
    "stray_bracket": "] render { }\n",
    "key_is_brace": "render { { } }\n",
    "comment_then_error": "# a comment line\nmaterial { colour [1 1 1] }\n",
}


def hexf(a):
    return [float(x).hex() for x in np.asarray(a, np.float32).ravel()]


def ref_binary_render(text, want_image=True):
    """Runs the reference executable on a scene text; returns (md5, bmp bytes)."""
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "scene.txt")
        with open(path, "w") as f:
            f.write(text)
        subprocess.run([orc.REF_BIN, path], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        with open(os.path.join(tmp, "image.bmp"), "rb") as f:
            data = f.read()
    return hashlib.md5(data).hexdigest(), data


def ray_fixture(name):
    path = os.path.join(orc.SCENES, name)
    orc.ref().ref_load(path.encode())
    info = orc.ref_info()
    osc = orc.Scene.load(path).build_bvh()      # only for the camera and the root box
    nodes = orc.ref_array("ref_nodes", orc.NODE_DTYPE, info["nodes"])
    lo, hi = nodes["lo"][0], nodes["hi"][0]
    pad = (hi - lo) * 0.25
    rr = np.concatenate([
        raygen.camera_grid(osc.camera(), info["width"], info["height"], step=6),
        raygen.random_rays(4096, lo - pad, hi + pad, seed=2024),
        raygen.axis_rays(),
    ]).astype(np.float32)
    n = rr.shape[0]
    idx = np.empty(n, np.int32)
    t = np.empty(n, np.float32)
    u = np.empty(n, np.float32)
    v = np.empty(n, np.float32)
    point = np.empty((n, 3), np.float32)
    normal = np.empty((n, 3), np.float32)
    orc.ref().ref_nearest(rr, n, idx, t, u, v, point, normal)
    rng = np.random.default_rng(5)
    dist = (rng.random(n, dtype=np.float32) * np.float32(4)).astype(np.float32)
    occ = np.empty(n, np.uint8)
    orc.ref().ref_occluded(rr, dist, n, occ)
    spheres = orc.ref_array("ref_objects", orc.OBJ_DTYPE, info["objects"])["type"][np.maximum(idx, 0)] == 1
    u[spheres | (idx < 0)] = 0          # barycentrics are only defined for triangle hits
    v[spheres | (idx < 0)] = 0
    np.savez_compressed(os.path.join(HERE, f"rays_{name.split('.')[0]}.npz"), rays=rr, idx=idx, t=t, u=u, v=v,
                        point=point, normal=normal, dist=dist, occluded=occ)
    return info, nodes


def main():
    if not orc.have_ref():
        sys.exit("build the reference first: make -C oracle ref")
    out = {"scenes": {}, "renders": {}, "small_renders": {}, "tolerance_renders": {}, "kat": {}, "parse_errors": {}}

    for name in SCENES:
        info, nodes = ray_fixture(name)
        objs = orc.ref_array("ref_objects", orc.OBJ_DTYPE, info["objects"])
        mats = orc.ref_array("ref_materials", orc.MAT_DTYPE, info["materials"])
        lights = orc.ref_array("ref_lights", orc.LIGHT_DTYPE, info["lights"])
        out["scenes"][name] = {
            "info": info,
            "nodes_md5": hashlib.md5(nodes.tobytes()).hexdigest(),
            "objects_md5": hashlib.md5(objs.tobytes()).hexdigest(),
            "materials_md5": hashlib.md5(mats.tobytes()).hexdigest(),
            "lights_md5": hashlib.md5(lights.tobytes()).hexdigest(),
        }
        if name == "cornell_box.txt":
            np.savez_compressed(os.path.join(HERE, "bvh_cornell.npz"), nodes=nodes, objects=objs, materials=mats,
                                lights=lights)

        # whole-path known answers: RayTrace from chosen rays and rng states
        osc = orc.Scene.load(os.path.join(orc.SCENES, name)).build_bvh()
        cam_rays = raygen.camera_grid(osc.camera(), info["width"], info["height"], step=50)
        paths = []
        for k, ray in enumerate(cam_rays):
            st = C.c_uint32(69420 + 977 * k)
            rad = np.zeros(3, np.float32)
            orc.ref().ref_trace_path(np.ascontiguousarray(ray), C.byref(st), rad)
            paths.append({"ray": hexf(ray), "seed": 69420 + 977 * k, "radiance": hexf(rad), "state_after": st.value})
        out["kat"][f"paths_{name}"] = paths

    # full-size renders of the reference binary: md5 only
    out["renders"]["cornell_box.txt default"] = ref_binary_render(orc.scene_variant("cornell_box.txt"))[0]
    out["renders"]["dragon.txt default"] = ref_binary_render(orc.scene_variant("dragon.txt"))[0]
    out["renders"]["dragon.txt 1920x1080x32"] = ref_binary_render(orc.scene_variant("dragon.txt", 1920, 1080, 32))[0]
    out["renders"]["cornell_box.txt 1920x1080x16"] = ref_binary_render(
        orc.scene_variant("cornell_box.txt", 1920, 1080, 16))[0]
    # small renders: the BMP itself
    for name, (w, h, spp) in {"cornell_box.txt": (64, 36, 8), "dragon.txt": (64, 36, 8),
                              "cornell_box.txt#odd": (33, 17, 3)}.items():
        scene = name.split("#")[0]
        md5, data = ref_binary_render(orc.scene_variant(scene, w, h, spp))
        fname = f"ref_{scene.split('.')[0]}_{w}x{h}x{spp}.bmp"
        with open(os.path.join(HERE, fname), "wb") as f:
            f.write(data)
        out["small_renders"][fname] = {"scene": scene, "width": w, "height": h, "samples": spp, "md5": md5}

    # renders with enough samples for a stated-tolerance comparison of images rendered from
    # other random streams (the reference's own streams are per image row)
    out["tolerance_renders"] = {}
    for scene, (w, h, spp) in {"cornell_box.txt": (200, 200, 256), "dragon.txt": (200, 200, 256)}.items():
        md5, data = ref_binary_render(orc.scene_variant(scene, w, h, spp))
        fname = f"ref_{scene.split('.')[0]}_{w}x{h}x{spp}.bmp"
        with open(os.path.join(HERE, fname), "wb") as f:
            f.write(data)
        out["tolerance_renders"][fname] = {"scene": scene, "width": w, "height": h, "samples": spp, "md5": md5}

    # random numbers
    st = C.c_uint32(69420)
    out["kat"]["rng_u32"] = [int(orc.ref().ref_rng_next(C.byref(st))) for _ in range(8)]
    st = C.c_uint32(69420)
    out["kat"]["rng_float"] = [float(orc.ref().ref_rng_float(C.byref(st))).hex() for _ in range(8)]
    tri = []
    st = C.c_uint32(12345)
    for _ in range(6):
        b = np.zeros(3, np.float32)
        orc.ref().ref_random_triangle(C.byref(st), b)
        tri.append(hexf(b))
    out["kat"]["random_triangle"] = {"seed": 12345, "values": tri, "state_after": st.value}

    # BSDF: evaluation, pdf and sampling over a material sweep
    rng = np.random.default_rng(99)
    mats = [(0.725, 0.71, 0.68, 1.0, 1.5, 0.0), (0.63, 0.065, 0.05, 0.01, 1.5, 0.0), (0.9, 0.6, 0.2, 0.09, 1.5, 1.0),
            (0.2, 0.4, 0.8, 0.25, 1.33, 0.5), (1.0, 1.0, 1.0, 0.0025, 2.4, 0.3)]
    bsdf = []
    for m in mats:
        for k in range(12):
            v = rng.standard_normal(3).astype(np.float32)
            v[2] = abs(v[2]) + np.float32(0.05)
            v /= np.linalg.norm(v)
            l = rng.standard_normal(3).astype(np.float32)
            if k % 4:
                l[2] = abs(l[2])
            l /= np.linalg.norm(l)
            ev = np.zeros(4, np.float32)
            orc.ref().ref_eval_bsdf(v, l, np.float32(m), ev)
            seed = int(rng.integers(1, 2 ** 32 - 1))
            st = C.c_uint32(seed)
            sm = np.zeros(7, np.float32)
            orc.ref().ref_sample_bsdf(v, np.float32(m), C.byref(st), sm)
            bsdf.append({"mat": hexf(m), "v": hexf(v), "l": hexf(l), "eval": hexf(ev), "seed": seed,
                         "sample": hexf(sm), "state_after": st.value})
    out["kat"]["bsdf"] = bsdf

    tone = []
    for c in (0.0, 1e-4, 0.0031308, 0.0032, 0.01, 0.18, 0.5, 1.0, 2.5, 17.0, 1e6):
        for e in (1.0, 0.5, 2.0):
            tone.append({"c": float(np.float32(c)).hex(), "exposure": e,
                         "out": float(orc.ref().ref_tonemap(np.float32(c), e)).hex()})
    out["kat"]["tonemap"] = tone

    # parse errors: the reference binary's stderr and exit status
    with tempfile.TemporaryDirectory() as tmp:
        for key, text in BAD_SCENES.items():
            path = os.path.join(tmp, "bad.txt")
            with open(path, "w") as f:
                f.write(text)
            p = subprocess.run([orc.REF_BIN, "bad.txt"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                               timeout=20)
            out["parse_errors"][key] = {"text": text, "status": p.returncode,
                                        "stderr": p.stderr.decode(errors="replace").strip()}
        p = subprocess.run([orc.REF_BIN], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        out["parse_errors"]["#usage"] = {"status": p.returncode, "stderr": p.stderr.decode().strip()}
        p = subprocess.run([orc.REF_BIN, "missing.txt"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        out["parse_errors"]["#missing_file"] = {"status": p.returncode, "stderr": p.stderr.decode().strip()}

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
