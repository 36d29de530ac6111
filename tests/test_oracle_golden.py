"""The CPU checker against the reference's own answers.

The fixtures under tests/golden/ were produced by the real reference compiled
from its own sources (tests/golden/make_golden.py); they pin the checker on
machines where the reference is not available.  Bars: everything bit-exact.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "golden.json")) as f:
    G = json.load(f)


def unhex(values):
    return np.array([float.fromhex(x) for x in values], np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def load(name):
    return orc.Scene.load(os.path.join(orc.SCENES, name)).build_bvh()


@pytest.mark.parametrize("name", ["cornell_box.txt", "dragon.txt"])
def test_scene_and_bvh_match_reference(name):
    sc = load(name)
    want = G["scenes"][name]
    info = sc.info()
    for key in ("objects", "nodes", "materials", "lights", "width", "height", "samples"):
        assert info[key] == want["info"][key], key
    assert hashlib.md5(sc.nodes().tobytes()).hexdigest() == want["nodes_md5"]
    assert hashlib.md5(sc.objects().tobytes()).hexdigest() == want["objects_md5"]
    assert hashlib.md5(sc.materials().tobytes()).hexdigest() == want["materials_md5"]
    assert hashlib.md5(sc.lights().tobytes()).hexdigest() == want["lights_md5"]


def test_cornell_tree_shape():
    z = np.load(os.path.join(GOLD, "bvh_cornell.npz"))
    sc = load("cornell_box.txt")
    assert sc.nodes().tobytes() == z["nodes"].tobytes()
    assert sc.objects().tobytes() == z["objects"].tobytes()
    nodes = sc.nodes()
    # root plus two leaves of 7 and 16 objects: the midpoint split fails at once on the wall triangles
    assert len(nodes) == 3 and nodes["count"].tolist() == [0, 7, 16]
    assert sc.depth() == 1


def test_dragon_tree_statistics():
    sc = load("dragon.txt")
    nodes = sc.nodes()
    leaves = nodes["count"][nodes["count"] > 0]
    assert len(nodes) == 12565 and len(leaves) == 6283 and leaves.max() == 7
    assert np.bincount(leaves)[1:].tolist() == [2251, 3804, 189, 26, 9, 3, 1]
    assert sc.depth() == 19


@pytest.mark.parametrize("name", ["cornell_box.txt", "dragon.txt"])
def test_ray_batches_match_reference(name):
    z = np.load(os.path.join(GOLD, f"rays_{name.split('.')[0]}.npz"))
    sc = load(name)
    idx, t, u, v = sc.trace_closest(z["rays"])
    assert np.array_equal(idx, z["idx"])
    assert np.array_equal(bits(t), bits(z["t"]))
    assert np.array_equal(bits(u), bits(z["u"]))
    assert np.array_equal(bits(v), bits(z["v"]))
    hit = idx >= 0
    point, normal = sc.hit_surface(z["rays"], idx, t, u, v)
    assert np.array_equal(bits(point[hit]), bits(z["point"][hit]))
    assert np.array_equal(bits(normal[hit]), bits(z["normal"][hit]))
    assert np.array_equal(sc.trace_occluded(z["rays"], z["dist"]), z["occluded"])
    assert 0.05 < hit.mean() < 1.0


def test_known_nearest_hits():
    """Pixel-centre camera rays at 400x400: object index and distance."""
    import rays as raygen
    cases = {
        "cornell_box.txt": {(200, 200): (8, "0x1.fa9c04p+0"), (100, 300): (0, "0x1.37f9e8p+1"),
                            (0, 0): (2, "0x1.b7568ap+0"), (399, 399): (15, "0x1.bbc694p+0"),
                            (250, 300): (5, "0x1.74785ep+0")},
        "dragon.txt": {(200, 200): (2719, "0x1.e525f2p+0"), (100, 300): (-1, None), (0, 0): (-1, None),
                       (399, 399): (-1, None)},
    }
    for name, want in cases.items():
        sc = load(name)
        grid = raygen.camera_grid(sc.camera(), 400, 400)
        for (u, v), (obj, t_hex) in want.items():
            idx, t, _, _ = sc.trace_closest(grid[v * 400 + u][None])
            assert idx[0] == obj, (name, u, v)
            if t_hex:
                assert float(t[0]) == float.fromhex(t_hex), (name, u, v)


def test_rng_known_answers():
    st = C.c_uint32(69420)
    assert [orc.lib().orc_rng_next(C.byref(st)) for _ in range(8)] == G["kat"]["rng_u32"]
    assert G["kat"]["rng_u32"][:4] == [494363678, 3480174692, 3407456629, 953793454]
    st = C.c_uint32(69420)
    got = [float(orc.lib().orc_rng_float(C.byref(st))).hex() for _ in range(8)]
    assert got == G["kat"]["rng_float"]


def test_stream_seed_never_zero_and_spread():
    seeds = {orc.lib().orc_stream_seed(69420, p, b) for p in range(2000) for b in range(8)}
    assert 0 not in seeds and len(seeds) == 16000


def test_bsdf_known_answers():
    for case in G["kat"]["bsdf"]:
        m, v, l = unhex(case["mat"]), unhex(case["v"]), unhex(case["l"])
        assert np.array_equal(bits(orc.eval_bsdf(v, l, m)), bits(unhex(case["eval"])))
        got, state = orc.sample_bsdf(v, m, case["seed"])
        assert np.array_equal(bits(got), bits(unhex(case["sample"])))
        assert state == case["state_after"]


def test_tonemap_known_answers():
    for case in G["kat"]["tonemap"]:
        got = orc.lib().orc_tonemap(orc.MATH_LIBM, float.fromhex(case["c"]), case["exposure"])
        assert float(np.float32(got)).hex() == case["out"]


@pytest.mark.parametrize("name", ["cornell_box.txt", "dragon.txt"])
def test_whole_paths_match_reference(name):
    sc = load(name)
    lit = 0
    for case in G["kat"][f"paths_{name}"]:
        rad, state = sc.trace_path(unhex(case["ray"]), case["seed"])
        assert np.array_equal(bits(rad), bits(unhex(case["radiance"])))
        assert state == case["state_after"]
        lit += bool(rad.any())
    assert lit > 10


@pytest.mark.parametrize("fname", sorted(G["small_renders"]))
def test_small_renders_match_reference_bmp(fname):
    meta = G["small_renders"][fname]
    sc = load(meta["scene"]).set_render(meta["width"], meta["height"], meta["samples"])
    bgr, _, _ = sc.render(orc.RNG_ROW, orc.MATH_LIBM)
    data = orc.bmp_bytes(bgr)
    with open(os.path.join(GOLD, fname), "rb") as f:
        want = f.read()
    assert hashlib.md5(want).hexdigest() == meta["md5"]
    assert data == want


@pytest.mark.parametrize("key", sorted(G["renders"]))
def test_full_renders_match_reference_md5(key):
    """The four renders whose image.bmp md5 the reference binary produced here."""
    scene, spec = key.split(" ")
    sc = load(scene)
    if spec != "default":
        w, h, spp = (int(x) for x in spec.split("x"))
        sc.set_render(w, h, spp)
    bgr, _, cn = sc.render(orc.RNG_ROW, orc.MATH_LIBM)
    assert hashlib.md5(orc.bmp_bytes(bgr)).hexdigest() == G["renders"][key]
    i = sc.info()
    assert cn["samples"] == i["width"] * i["height"] * i["samples"]


def test_render_is_thread_count_independent():
    sc = load("cornell_box.txt").set_render(48, 32, 4)
    a = sc.render(orc.RNG_ROW, orc.MATH_LIBM, threads=1)[1]
    b = sc.render(orc.RNG_ROW, orc.MATH_LIBM, threads=5)[1]
    assert np.array_equal(bits(a), bits(b))
    a = sc.render(orc.RNG_STREAM, orc.MATH_PM, seed=3, samples_per_stream=3, threads=1)[1]
    b = sc.render(orc.RNG_STREAM, orc.MATH_PM, seed=3, samples_per_stream=3, threads=7)[1]
    assert np.array_equal(bits(a), bits(b))


def test_stream_mode_is_statistically_the_same_image():
    """Per-(pixel, block) streams and the portable elementary functions change
    the noise, not the image: RMSE against the reference-identical render stays
    at the Monte-Carlo noise level, and the two math modes agree far below it."""
    sc = load("dragon.txt").set_render(100, 100, 64)
    ref = sc.render(orc.RNG_ROW, orc.MATH_LIBM)[0].astype(np.float64)
    other_seed = sc.render(orc.RNG_STREAM, orc.MATH_LIBM, seed=1, samples_per_stream=16)[0].astype(np.float64)
    pm = sc.render(orc.RNG_STREAM, orc.MATH_PM, seed=1, samples_per_stream=16)[0].astype(np.float64)
    rmse = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
    noise = rmse(ref, other_seed)
    assert noise < 4.0                      # 8-bit units; ~2.3 measured at 64 spp
    assert rmse(other_seed, pm) < noise      # same streams, only last-bit math differences
    assert abs(ref.mean() - pm.mean()) < 0.5


@pytest.mark.parametrize("fname,floor", [("ref_dragon_200x200x256.bmp", 1.13), ("ref_cornell_box_200x200x256.bmp", 3.31)])
def test_stream_mode_within_stated_tolerance_of_reference_bmp(fname, floor):
    """The image tolerance the GPU tests apply, applied to the checker itself:
    against the 200x200 x 256 spp image the reference binary rendered, a render
    from per-(pixel, block) streams with the portable elementary functions has
    RMSE <= 1.5 x the reference's own seed-to-seed noise floor at 256 spp and a
    mean difference per channel of at most 0.3 / 255."""
    meta = G["tolerance_renders"][fname]
    with open(os.path.join(GOLD, fname), "rb") as f:
        data = f.read()
    assert hashlib.md5(data).hexdigest() == meta["md5"]
    want = orc.read_bmp(data).astype(np.float64)
    sc = load(meta["scene"]).set_render(meta["width"], meta["height"], meta["samples"])
    got = sc.render(orc.RNG_STREAM, orc.MATH_PM, seed=69420, samples_per_stream=16)[0].astype(np.float64)
    d = got - want
    assert float(np.sqrt((d ** 2).mean())) <= 1.5 * floor
    assert (np.abs(d.reshape(-1, 3).mean(axis=0)) <= 0.3).all()
