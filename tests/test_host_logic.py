"""The shipped library's host side (scene grammar, BVH build, camera, BMP,
sharding, validation) against the CPU checker and the reference's fixtures.
No GPU needed.
"""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import orc
import raytracer_amd as rta

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "golden.json")) as f:
    G = json.load(f)


def pair(name):
    path = os.path.join(orc.SCENES, name)
    return orc.Scene.load(path).build_bvh(), rta.Scene.load(path).build_bvh()


@pytest.mark.parametrize("name", ["cornell_box.txt", "dragon.txt"])
def test_flat_arrays_equal_checker(name):
    osc, psc = pair(name)
    a, ob, on = psc.arrays(), osc.objects(), osc.nodes()
    assert np.array_equal(a["prim_type"], ob["type"])
    assert np.array_equal(a["prim_material"], ob["material"])
    assert np.array_equal(a["prim_light"], ob["light"])
    assert a["prim_geom"].tobytes() == ob["geom"][:, :9].tobytes()
    tri = ob["type"] == 0
    assert a["prim_normals"][tri].tobytes() == ob["geom"][tri][:, 9:].tobytes()
    assert a["node_bounds"].tobytes() == np.concatenate([on["lo"], on["hi"]], axis=1).tobytes()
    assert np.array_equal(a["node_index"], on["index"]) and np.array_equal(a["node_count"], on["count"])
    assert a["materials"].tobytes() == osc.materials().tobytes()
    assert np.array_equal(a["light_prim"], osc.lights()["object"])
    assert a["light_color"].tobytes() == osc.lights()["color"].tobytes()
    rta.validate(psc.desc())


def test_permutation_maps_back_to_file_order():
    path = os.path.join(orc.SCENES, "dragon.txt")
    unbuilt = orc.Scene.load(path)          # file order
    psc = rta.Scene.load(path).build_bvh()
    perm = psc.permutation()
    assert sorted(perm.tolist()) == list(range(len(perm)))
    assert psc.arrays()["prim_geom"].tobytes() == unbuilt.objects()["geom"][perm][:, :9].tobytes()


def test_camera_and_settings():
    for name in ("cornell_box.txt", "dragon.txt"):
        osc, psc = pair(name)
        cam = psc.camera()
        got = np.concatenate([np.array(getattr(cam, k), np.float32) for k in
                              ("origin", "upper_left", "du", "dv", "lens_u", "lens_v")] +
                             [np.float32([cam.defocus_angle])])
        assert got.tobytes() == osc.camera().tobytes()
        s, i, p = psc.settings, osc.info(), osc.params()
        assert (s.width, s.height, s.samples) == (i["width"], i["height"], i["samples"])
        assert list(s.sky) == list(p["sky"]) and s.exposure == p["exposure"] and s.fov == p["fov"]


def test_defaults_of_an_empty_render_block():
    sc = rta.Scene.parse("sphere { radius 1 center [0 0 3] }")
    s = sc.settings
    assert (s.width, s.height, s.samples, s.fov, s.exposure, s.defocus_angle) == (512, 512, 20, 90.0, 1.0, -1.0)
    assert list(s.up) == [0, 0, 1] and list(s.sky) == [0, 0, 0]
    a = sc.build_bvh().arrays()
    assert a["materials"].tolist() == [[0.5, 0.5, 0.5, 1.0, 1.5, 0.0]]      # material 0, src/parser.cpp:255-258
    assert a["prim_type"].tolist() == [rta.PRIM_SPHERE] and a["node_count"].tolist() == [1]


def test_sticky_state_and_squared_roughness():
    text = """
    triangle_mesh { vertices [ 0 0 0 1 0 0 0 1 0 ] }          # default material 0
    material { color [1 0 0] roughness .5 }
    triangle_mesh { vertices [ 0 0 1 1 0 1 0 1 1   0 0 2 1 0 2 0 1 2 ] }
    area_light { color [3 2 1] }
    sphere { radius 1 center [5 5 5] }                        # spheres never emit
    triangle_mesh { vertices [ 0 0 3 1 0 3 0 1 3 ] }
    material { }                                               # defaults .7 grey
    triangle_mesh { vertices [ 0 0 4 1 0 4 0 1 4 ] }          # still a light
    """
    sc = rta.Scene.parse(text)
    osc = orc.Scene.parse(text)
    raw = osc.objects()
    assert raw["material"].tolist() == [0, 1, 1, 1, 1, 2]
    assert raw["light"].tolist() == [-1, -1, -1, -1, 0, 1]
    built = sc.build_bvh().arrays()
    mats = built["materials"]
    assert mats[1].tolist() == [1, 0, 0, 0.25, 1.5, 0] and np.allclose(mats[2], [.7, .7, .7, 1, 1.5, 0])
    assert built["light_color"].tolist() == [[3, 2, 1], [3, 2, 1]]
    for li, p in enumerate(built["light_prim"]):
        assert built["prim_light"][p] == li


@pytest.mark.parametrize("key", sorted(k for k in G["parse_errors"] if not k.startswith("#")))
def test_parse_errors_match_reference(key):
    case = G["parse_errors"][key]
    with pytest.raises(rta.RtaError) as e:
        rta.Scene.parse(case["text"], name="bad.txt")
    assert e.value.code == rta.E_PARSE
    assert e.value.text == case["stderr"]
    with pytest.raises(ValueError) as e2:
        orc.Scene.parse(case["text"], name="bad.txt")
    assert str(e2.value) == case["stderr"]


def test_missing_file():
    with pytest.raises(rta.RtaError) as e:
        rta.Scene.load("/nonexistent/scene.txt")
    assert e.value.code == rta.E_IO and e.value.text == "Couldn't open '/nonexistent/scene.txt'"
    assert G["parse_errors"]["#missing_file"]["stderr"].endswith("Couldn't open 'missing.txt'")


def test_comments_whitespace_and_crlf():
    a = rta.Scene.parse("render{width 8 height 4}#x\nsphere{radius 1 center[0 0 0]}").settings
    b = rta.Scene.parse("render {\r\n\twidth 8\r\n\theight 4 # c\r\n}\r\nsphere { radius 1 center [ 0 0 0 ] }\r\n").settings
    assert (a.width, a.height) == (b.width, b.height) == (8, 4)
    assert rta.Scene.parse("# only a comment, no newline").settings.width == 512


def test_bmp_writer_bytes():
    rng = np.random.default_rng(0)
    for w, h in ((1, 1), (3, 2), (4, 4), (33, 17), (64, 36)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "x.bmp")
            rta.write_bmp(path, img)
            with open(path, "rb") as f:
                data = f.read()
        assert data == orc.bmp_bytes(img)
        assert len(data) == 54 + h * ((w * 3 + 3) // 4 * 4)
        assert np.array_equal(orc.read_bmp(data), img)
    # and against a file the reference binary wrote
    with open(os.path.join(GOLD, "ref_cornell_box_33x17x3.bmp"), "rb") as f:
        want = f.read()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "y.bmp")
        rta.write_bmp(path, orc.read_bmp(want))
        with open(path, "rb") as f:
            assert f.read() == want


@pytest.mark.parametrize("w,h,n", [(64, 36, 1), (64, 36, 3), (33, 17, 2), (1920, 1080, 8), (5, 5, 8)])
def test_shards_partition_the_frame(w, h, n):
    seen = np.zeros(w * h, np.int32)
    sizes = []
    for r in range(n):
        px = rta.shard_pixels(w, h, r, n)
        seen[px] += 1
        sizes.append(len(px))
        if len(px):
            # walk order: 8x8 tiles, so the first pixels of a shard sit in one tile
            first = px[: min(64, len(px))]
            assert (first // w).max() - (first // w).min() < 8 and (first % w).max() - (first % w).min() < 8
    assert (seen == 1).all()
    if w * h > 64 * n * 4:
        assert max(sizes) - min(sizes) <= 64 * (1 + (w * h // 64) // (n * 50))
    with pytest.raises(rta.RtaError):
        rta.shard_pixels(w, h, n, n)


def test_validation_rejects_malformed_descriptions():
    path = os.path.join(orc.SCENES, "cornell_box.txt")
    good = rta.Scene.load(path).build_bvh().arrays()
    rta.validate(rta.desc_from_arrays(good))

    def broken(**changes):
        a = {k: v.copy() for k, v in good.items()}
        for k, (i, val) in changes.items():
            a[k][i] = val
        return rta.desc_from_arrays(a)

    for bad, word in [
        (broken(prim_material=(0, 99)), "material"),
        (broken(prim_light=(0, 7)), "light"),
        (broken(prim_type=(0, 5)), "type"),
        (broken(light_prim=(0, 10 ** 6)), "primitive"),
        (broken(node_index=(0, 0)), "child"),              # root pointing at itself
        (broken(node_index=(0, 2)), "child"),              # second child past the array
        (broken(node_count=(1, 99)), "range"),             # leaf overruns the primitives
        (broken(node_count=(0, -1)), "negative"),
        (broken(node_index=(2, 0)), "two leaves"),         # overlapping leaves
    ]:
        with pytest.raises(rta.RtaError) as e:
            rta.validate(bad)
        assert word in e.value.text, e.value.text


def test_deep_chain_is_rejected():
    # a left-deep tree of 70 levels: deeper than the 64-entry traversal stack
    levels = 70
    n_nodes = 2 * levels + 1
    idx = np.zeros(n_nodes, np.int32)
    cnt = np.zeros(n_nodes, np.int32)
    for d in range(levels):
        node = 0 if d == 0 else 2 * d - 1
        idx[node] = 2 * d + 1
        cnt[2 * d + 2] = 1
        idx[2 * d + 2] = d
    cnt[2 * levels - 1] = 1
    idx[2 * levels - 1] = levels
    n_prims = levels + 1
    a = {"prim_type": np.zeros(n_prims, np.int32), "prim_material": np.zeros(n_prims, np.int32),
         "prim_light": -np.ones(n_prims, np.int32), "prim_geom": np.zeros((n_prims, 9), np.float32),
         "prim_normals": np.zeros((n_prims, 9), np.float32), "node_bounds": np.zeros((n_nodes, 6), np.float32),
         "node_index": idx, "node_count": cnt, "materials": np.ones((1, 6), np.float32),
         "light_color": np.zeros((0, 3), np.float32), "light_prim": np.zeros(0, np.int32)}
    with pytest.raises(rta.RtaError) as e:
        rta.validate(rta.desc_from_arrays(a))
    assert "depth" in e.value.text
    assert e.value.code == rta.E_LIMIT        # a size limit of the kernels, not a malformed description


def test_degenerate_input_builds_a_tree_the_kernels_cannot_take():
    """The reference builder bounds the pending second children, not the depth:
    centroids that halve the remaining range again and again give a chain deeper
    than 64 levels.  The host builder must reproduce it (as the checker does)
    and validation must name the limit."""
    n = 120
    tri = np.zeros((n, 9), np.float32)
    for k in range(n):
        x = -np.float32(2.0) ** np.float32(-k)      # -1, -1/2, -1/4, ...: a midpoint split peels one or two triangles off the
        #                                              FIRST child's side, so the chain runs down the second children
        tri[k] = [x, 0, 0, x, 1e-44, 0, x, 0, 1e-44]
    p = rta.Scene.from_triangles(tri).build_bvh()
    o = orc.Scene.from_triangles(tri).build_bvh()
    a = p.arrays()
    assert np.array_equal(a["node_index"], o.nodes()["index"]) and np.array_equal(a["node_count"], o.nodes()["count"])
    assert o.depth() > 64
    with pytest.raises(rta.RtaError) as e:
        rta.validate(p.desc())
    assert e.value.code == rta.E_LIMIT and "depth" in e.value.text


def test_inverted_boxes_are_accepted():
    """min > max on an axis is legal in a caller-made tree (the reference's slab
    test selects planes by the direction's sign bit whatever their order)."""
    path = os.path.join(orc.SCENES, "cornell_box.txt")
    a = rta.Scene.load(path).build_bvh().arrays()
    a["node_bounds"] = a["node_bounds"].copy()
    a["node_bounds"][1, [0, 3]] = a["node_bounds"][1, [3, 0]]
    rta.validate(rta.desc_from_arrays(a))


def test_synthetic_soup_bvh_equals_checker():
    rng = np.random.default_rng(1)
    n = 20000
    c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
    tri = (c + rng.uniform(-0.01, 0.01, (n, 3, 3)).astype(np.float32)).reshape(n, 9)
    o = orc.Scene.from_triangles(tri).build_bvh()
    p = rta.Scene.from_triangles(tri).build_bvh()
    a = p.arrays()
    assert a["prim_geom"].tobytes() == o.objects()["geom"][:, :9].tobytes()
    assert np.array_equal(a["node_index"], o.nodes()["index"])
    assert a["node_bounds"].tobytes() == np.concatenate([o.nodes()["lo"], o.nodes()["hi"]], axis=1).tobytes()
    rta.validate(p.desc())


def test_build_requires_primitives_and_happens_once():
    with pytest.raises(rta.RtaError):
        rta.Scene.parse("render { width 4 }").build_bvh()
    sc = rta.Scene.parse("sphere { radius 1 center [0 0 0] }").build_bvh()
    with pytest.raises(rta.RtaError):
        sc.build_bvh()
