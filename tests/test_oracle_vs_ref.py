"""The CPU checker against the real reference, compiled from its own sources
into oracle/_ref/ (make -C oracle).  Only runs where that build exists, i.e.
in the build container; the same answers travel as fixtures (test_oracle_golden).
"""
import ctypes as C
import hashlib
import os
import subprocess
import tempfile

import numpy as np
import pytest

import orc
import rays as raygen

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", ["cornell_box.txt", "dragon.txt"])
def test_arrays_and_rays(name):
    path = os.path.join(orc.SCENES, name)
    orc.ref().ref_load(path.encode())
    info = orc.ref_info()
    sc = orc.Scene.load(path).build_bvh()
    assert sc.objects().tobytes() == orc.ref_array("ref_objects", orc.OBJ_DTYPE, info["objects"]).tobytes()
    assert sc.nodes().tobytes() == orc.ref_array("ref_nodes", orc.NODE_DTYPE, info["nodes"]).tobytes()
    assert sc.lights().tobytes() == orc.ref_array("ref_lights", orc.LIGHT_DTYPE, info["lights"]).tobytes()
    nodes = sc.nodes()
    rr = np.concatenate([raygen.camera_grid(sc.camera(), 400, 400, step=3),
                         raygen.random_rays(30000, nodes["lo"][0] - 0.5, nodes["hi"][0] + 0.5, seed=77),
                         raygen.axis_rays()])
    n = rr.shape[0]
    idx = np.empty(n, np.int32)
    t, u, v = (np.empty(n, np.float32) for _ in range(3))
    point, normal = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
    orc.ref().ref_nearest(rr, n, idx, t, u, v, point, normal)
    oi, ot, ou, ov = sc.trace_closest(rr)
    assert np.array_equal(oi, idx)
    assert np.array_equal(bits(ot), bits(t))
    tri = (idx >= 0) & (sc.objects()["type"][np.maximum(idx, 0)] == 0)
    assert np.array_equal(bits(ou[tri]), bits(u[tri])) and np.array_equal(bits(ov[tri]), bits(v[tri]))
    dist = (np.random.default_rng(3).random(n, dtype=np.float32) * 4).astype(np.float32)
    occ = np.empty(n, np.uint8)
    orc.ref().ref_occluded(rr, dist, n, occ)
    assert np.array_equal(sc.trace_occluded(rr, dist), occ)


def test_reference_binary_image_is_reproduced():
    text = orc.scene_variant("cornell_box.txt", 120, 90, 6)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "s.txt")
        with open(path, "w") as f:
            f.write(text)
        subprocess.run([orc.REF_BIN, path], cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        with open(os.path.join(tmp, "image.bmp"), "rb") as f:
            want = f.read()
    bgr, _, _ = orc.Scene.parse(text).build_bvh().render(orc.RNG_ROW, orc.MATH_LIBM)
    assert orc.bmp_bytes(bgr) == want


def test_defocus_and_exposure_paths():
    """Features no shipped scene uses: the lens sampler with its x*y < 1 accept
    test, exposure, a metallic rough material, a second sphere."""
    text = orc.scene_variant("cornell_box.txt", 80, 60, 5).replace(
        "look_at [0 0 1]", "look_at [0 0 1]\n\tdefocus_angle 2.5\n\texposure 1.7\n\tfov 70")
    text += "\nmaterial { color [0.9 0.8 0.3] roughness 0.35 metallic 1 ior 1.8 }\nsphere { center [0.3 -0.4 0.25] radius 0.25 }\n"
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "s.txt")
        with open(path, "w") as f:
            f.write(text)
        subprocess.run([orc.REF_BIN, path], cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        with open(os.path.join(tmp, "image.bmp"), "rb") as f:
            want = f.read()
    bgr, _, _ = orc.Scene.parse(text).build_bvh().render(orc.RNG_ROW, orc.MATH_LIBM)
    assert orc.bmp_bytes(bgr) == want


def test_random_triangle_and_paths():
    st_a, st_b = C.c_uint32(4242), C.c_uint32(4242)
    for _ in range(200):
        assert orc.ref().ref_rng_next(C.byref(st_a)) == orc.lib().orc_rng_next(C.byref(st_b))
    path = os.path.join(orc.SCENES, "cornell_box.txt")
    orc.ref().ref_load(path.encode())
    sc = orc.Scene.load(path).build_bvh()
    rr = raygen.camera_grid(sc.camera(), 400, 400, step=23)
    for k, ray in enumerate(rr):
        st = C.c_uint32(1000 + k)
        want = np.zeros(3, np.float32)
        orc.ref().ref_trace_path(np.ascontiguousarray(ray), C.byref(st), want)
        got, state = sc.trace_path(ray, 1000 + k)
        assert np.array_equal(bits(got), bits(want)) and state == st.value
