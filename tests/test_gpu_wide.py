"""The certified walk over the wide tree (csrc/traverse_wide.hpp) against the
traversal that replays the reference's visit order (src/main.cpp:332-422), and
through it against the CPU checker: same primitive, t, u, v bits for every ray,
whether the walk proves its answer or hands the ray on."""
import numpy as np
import pytest

import orc
import rays as raygen
import raytracer_amd as rta

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _variant(gpu, variant, rr):
    import torch
    dev = torch.device("cuda", 0)
    rt = torch.from_numpy(np.ascontiguousarray(rr, np.float32)).to(dev)
    ht = torch.zeros((rr.shape[0], 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()      # the library works on its own streams: torch's fill and copy must be through
    gpu.trace_variant_device(variant, rt.data_ptr(), rr.shape[0], ht.data_ptr())
    return ht.cpu().numpy()


def walk(gpu, rr):
    """rta_debug_trace_variant 20: the certified walk, then the exact traversal for the rays left in doubt."""
    return _variant(gpu, 20, rr)


def exact(gpu, rr):
    """rta_debug_trace_variant 10: every ray through the traversal that replays the reference's order."""
    got = _variant(gpu, 10, rr)
    return dict(prim=got[:, 0].view(np.int32).copy(), t=got[:, 1].copy(), u=got[:, 2].copy(), v=got[:, 3].copy())


def batch(gpu, rr):
    """rta_trace_closest as callers use it (host buffers): the walk where the scene has a wide tree."""
    h = gpu.trace_closest(rr)
    return np.stack([h["prim"].view(np.float32), h["t"], h["u"], h["v"]], axis=1)


def same_hits(got, want):
    hit = want["prim"] >= 0
    ok = got[:, 0].view(np.int32) == want["prim"]
    ok &= got[:, 1].view(np.uint32) == bits(want["t"])
    ok &= ~hit | ((got[:, 2].view(np.uint32) == bits(want["u"])) & (got[:, 3].view(np.uint32) == bits(want["v"])))
    return ok


def _grid_mesh(n, z_of):
    """2 n^2 triangles over the unit square, heights from z_of(x, y); vertices on a binary grid, so rays
    aimed at them meet shared edges and corners exactly."""
    tris = []
    h = 1.0 / n
    for i in range(n):
        for j in range(n):
            p = [(i * h, j * h), ((i + 1) * h, j * h), ((i + 1) * h, (j + 1) * h), (i * h, (j + 1) * h)]
            q = [(x, y, z_of(x, y)) for x, y in p]
            tris.append(q[0] + q[1] + q[2])
            tris.append(q[0] + q[2] + q[3])
    return np.asarray(tris, np.float32)


def _mesh_scene(tris):
    body = " ".join("%r" % float(x) for x in tris.reshape(-1))
    return ("render {\n\tcamera [0.5 0.5 3]\n\tlook_at [0.5 0.5 0]\n\tup [0 1 0]\n\twidth 48\n\theight 48\n\tsamples 8\n\tfov 30\n"
            "\tsky_box_color [.5 .6 1]\n}\n\nmaterial {\n\tcolor [0.8 0.7 0.6]\n\troughness 1\n}\n\n"
            "triangle_mesh {\n\tvertices [ %s ]\n}\n" % body)


def test_walk_equals_the_order_replaying_traversal_on_the_dragon(gpu):
    osc = orc.Scene.load(f"{orc.SCENES}/dragon.txt").build_bvh()
    psc = rta.Scene.load(f"{orc.SCENES}/dragon.txt").build_bvh()
    gpu.upload(psc)
    nodes = osc.nodes()
    lo, hi = nodes["lo"][0], nodes["hi"][0]
    rng = np.random.default_rng(3)
    inside = raygen.random_rays(400000, lo, hi, seed=21)
    first = exact(gpu, inside)
    hit = first["prim"] >= 0
    # rays leaving the surface, as bounce rays do
    bounce = inside[hit].copy()
    bounce[:, 0:3] += bounce[:, 3:6] * first["t"][hit, None]
    d = rng.standard_normal(bounce[:, 3:6].shape).astype(np.float32)
    bounce[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    # rays aimed at vertices: they meet the triangles around the vertex at (nearly) the same distance
    geom = psc.arrays()["prim_geom"]
    verts = geom.reshape(-1, 3)[rng.integers(0, geom.shape[0] * 3, 100000)]
    o = (lo + (hi - lo) * rng.random((verts.shape[0], 3), dtype=np.float32)).astype(np.float32)
    d = (verts - o).astype(np.float32)
    aimed = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)], axis=1)
    # outside what the walk admits: far origins, long and tiny directions, zero components, NaN
    far = raygen.random_rays(20000, lo - 30, hi + 30, seed=5)
    long_dir = inside[:20000].copy()
    long_dir[:, 3:6] *= 5
    odd = np.concatenate([raygen.axis_rays(), np.float32([[0, 0, 1, np.nan, 0, 1], [np.nan, 0, 1, 0, 0, 1], [0, 0, 1, 1e-30, 1, 1e-30],
                                                         [0, 0, 1, 0, 0, 0]])])
    for label, rr in (("inside", inside), ("bounce", bounce), ("aimed at vertices", aimed), ("far", far), ("long", long_dir), ("odd", odd)):
        want = exact(gpu, rr)
        for how, got in (("probe", walk(gpu, rr)), ("rta_trace_closest", batch(gpu, rr))):
            ok = same_hits(got, want)
            assert ok.all(), f"{label} ({how}): {(~ok).sum()} rays differ, first {rr[np.nonzero(~ok)[0][0]].tolist()}"
    # occlusion queries take the any-hit walk: limits below, at and beyond the nearest hit, and the checker's answers
    for label, rr in (("inside", inside[:150000]), ("bounce", bounce[:150000]), ("aimed at vertices", aimed), ("far", far),
                      ("long", long_dir), ("odd", odd)):
        near = exact(gpu, rr)["t"]
        k = rng.integers(0, 6, rr.shape[0])
        scale = np.float32([0.5, 0.999, 1.0, 1.0000001, 1.5, 1e30])[k]
        with np.errstate(over="ignore"):
            limit = np.where(near < 3e38, near * scale, np.float32(10.0) * scale).astype(np.float32)
        limit[k == 5] = np.float32(3.4e38)
        want = osc.trace_occluded(rr, limit)
        got = gpu.trace_occluded(rr, limit)
        assert np.array_equal(got, want), f"occlusion, {label}: {(got != want).sum()} answers differ"
        if label in ("inside", "bounce", "aimed at vertices"):
            assert 0.1 < want.mean() < 0.9
    # and the exact traversal is the checker's (a sample; tests/test_gpu_parity.py covers it at length)
    sample = np.concatenate([bounce[:20000], aimed[:20000]])
    idx, t, u, v = osc.trace_closest(sample)
    want = dict(prim=idx, t=t, u=u, v=v)
    assert same_hits(walk(gpu, sample), want).all()


@pytest.mark.parametrize("shape", ["corner", "bumpy"])
def test_ties_and_flat_boxes_are_handed_on(gpu, shape):
    """A triangulated grid: rays through its vertices and along its edges hit two to six triangles at the same
    distance, and a flat grid's boxes (the floor and the wall of "corner") have no thickness, so the distance to a leaf's box equals the hit's.  The
    walk cannot prove such answers; the renders must still agree bit for bit with the checker, and report the
    rays that were handed on."""
    z_of = (lambda x, y: 0.0) if shape == "corner" else (lambda x, y: 0.25 * np.sin(7 * x) * np.cos(5 * y))
    tris = _grid_mesh(24, z_of)
    if shape == "corner":
        # a flat floor and a flat wall (the floor's grid stood up on the plane x = 0): paths bounce between them
        wall = tris.reshape(-1, 3, 3)[:, :, [2, 1, 0]]
        tris = np.concatenate([tris, wall.reshape(-1, 9)])
    text = _mesh_scene(tris)
    osc = orc.Scene.parse(text).build_bvh()
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    rng = np.random.default_rng(11)
    # from above and below, through grid vertices, edge midpoints and random points, straight and slanted
    k = rng.integers(0, 25, (60000, 2))
    target = np.stack([k[:, 0] / 24, k[:, 1] / 24], axis=1)
    target[20000:40000, 0] += 1 / 48                      # on an edge
    target[40000:] = rng.random((20000, 2))
    tz = np.array([z_of(x, y) for x, y in target], np.float64)
    o = np.stack([target[:, 0] + rng.choice([0, 0.25, -0.5], 60000), target[:, 1] + rng.choice([0, 0.125], 60000),
                  rng.choice([1.0, -1.0, 0.5], 60000)], axis=1)
    d = np.stack([target[:, 0], target[:, 1], tz], axis=1) - o
    rr = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], axis=1).astype(np.float32)
    idx, t, u, v = osc.trace_closest(rr)
    want = dict(prim=idx, t=t, u=u, v=v)
    for got in (walk(gpu, rr), batch(gpu, rr)):
        ok = same_hits(got, want)
        assert ok.all(), f"{(~ok).sum()} rays differ, first {rr[np.nonzero(~ok)[0][0]].tolist()}"
    assert (idx >= 0).sum() > 30000
    # occlusion along the same rays: limits at, just below and just beyond the hit (ties and flat boxes again)
    k = rng.integers(0, 5, rr.shape[0])
    with np.errstate(over="ignore"):
        limit = np.where(idx >= 0, t * np.float32([0.75, 0.9999999, 1.0, 1.0000001, 2.0])[k], np.float32(5.0)).astype(np.float32)
    assert np.array_equal(gpu.trace_occluded(rr, limit), osc.trace_occluded(rr, limit))
    # the same scene rendered: bounce rays start ON the grid
    want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=3, samples_per_stream=4)
    bgr, acc, st = gpu.render_scene(psc, seed=3, samples_per_stream=4)
    assert st["rays"] == cn["rays"]
    assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bgr, want_bgr)
    if shape == "corner":
        assert st["retraced_rays"] > 0


def test_render_with_and_without_the_wide_tree(gpu):
    text = orc.scene_variant("dragon.txt", 320, 180, 48)
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    bgr_a, acc_a, st_a = gpu.render_scene(psc, seed=9, samples_per_stream=8)
    bgr_b, acc_b, st_b = gpu.render_scene(psc, seed=9, samples_per_stream=8, flags=rta.RENDER_EXACT_ONLY)
    assert np.array_equal(bits(acc_a), bits(acc_b)) and np.array_equal(bgr_a, bgr_b)
    assert st_a["rays"] == st_b["rays"] and st_a["early_rays"] == st_b["early_rays"]
    assert st_b["retraced_rays"] == 0 and st_b["walked_rays"] == 0
    # the walk answers all but a sliver of the bounce rays itself
    bounce = st_a["rays"] - st_a["samples"]
    assert bounce > 100000 and st_a["retraced_rays"] < bounce // 1000
    # all bounce rays but those that need the division-based box test, and the camera rays where the camera is in reach
    assert bounce * 0.99 <= st_a["walked_rays"] <= st_a["rays"] - st_a["early_rays"]
    # dragon.txt's camera is a pinhole in reach: its rays are answered from the triangle lists of their cells
    # on the image plane, every one that is traced at all
    traced_camera = st_a["samples"] - st_a["early_rays"]
    assert 0.999 * traced_camera <= st_a["listed_rays"] <= traced_camera and st_b["listed_rays"] == 0
    assert st_a["walked_rays"] + st_a["listed_rays"] <= st_a["rays"] - st_a["early_rays"]


def test_a_few_spheres_beside_the_mesh(gpu):
    """Spheres stay outside the wide tree and are tested for every ray: a bumpy grid with a sphere above it, one cutting
    through it and a light; batches, occlusion queries and the render against the checker."""
    tris = _grid_mesh(20, lambda x, y: 0.1 * np.sin(9 * x) * np.cos(7 * y))
    body = " ".join("%r" % float(x) for x in tris.reshape(-1))
    text = ("render {\n\tcamera [0.5 0.5 2.2]\n\tlook_at [0.5 0.5 0]\n\tup [0 1 0]\n\twidth 56\n\theight 40\n\tsamples 16\n\tfov 30\n"
            "\tsky_box_color [.3 .4 .6]\n}\n\nmaterial {\n\tcolor [0.8 0.7 0.6]\n\troughness 0.4\n\tmetallic 1\n}\n\n"
            "sphere {\n\tcenter [0.3 0.6 0.35]\n\tradius 0.15\n}\n\nsphere {\n\tcenter [0.7 0.4 0.02]\n\tradius 0.12\n}\n\n"
            "material {\n\tcolor [0.7 0.8 0.6]\n\troughness 1\n}\n\ntriangle_mesh {\n\tvertices [ %s ]\n}\n\n"
            "material {\n\tcolor [0 0 0]\n\troughness 1\n}\n\narea_light {\n\tcolor [9 9 8]\n}\n\n"
            "triangle_mesh {\n\tvertices [ 0.4 0.4 1.0  0.6 0.4 1.0  0.5 0.6 1.05 ]\n}\n" % body)
    osc = orc.Scene.parse(text).build_bvh()
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    nodes = osc.nodes()
    lo, hi = nodes["lo"][0], nodes["hi"][0]
    rng = np.random.default_rng(17)
    rr = raygen.random_rays(200000, lo - 0.2, hi + 0.2, seed=9)
    # half of them aimed at the spheres
    for c, r in (((0.3, 0.6, 0.35), 0.15), ((0.7, 0.4, 0.02), 0.12)):
        sel = rng.random(rr.shape[0]) < 0.25
        target = np.float32(c) + rng.standard_normal((int(sel.sum()), 3)).astype(np.float32) * np.float32(r * 0.6)
        d = target - rr[sel, 0:3]
        rr[sel, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    idx, t, u, v = osc.trace_closest(rr)
    want = dict(prim=idx, t=t, u=u, v=v)
    for got in (walk(gpu, rr), batch(gpu, rr)):
        ok = same_hits(got, want)
        assert ok.all(), f"{(~ok).sum()} rays differ, first {rr[np.nonzero(~ok)[0][0]].tolist()}"
    types = psc.arrays()["prim_type"]
    assert (types[idx[idx >= 0]] != 0).sum() > 5000      # spheres are among the answers
    k = rng.integers(0, 5, rr.shape[0])
    with np.errstate(over="ignore"):
        limit = np.where(idx >= 0, t * np.float32([0.6, 0.9999999, 1.0, 1.0000001, 1.7])[k], np.float32(3.0)).astype(np.float32)
    assert np.array_equal(gpu.trace_occluded(rr, limit), osc.trace_occluded(rr, limit))
    want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=8, samples_per_stream=4)
    bgr, acc, st = gpu.render_scene(psc, seed=8, samples_per_stream=4)
    assert st["rays"] == cn["rays"] and st["shadow_rays"] == cn["shadow_rays"] and st["walked_rays"] > 0
    assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bgr, want_bgr)


def test_forty_spheres_beside_the_mesh(gpu):
    """Up to 48 spheres stay beside the wide tree (every ray tests each of them, from scalar operands); one more and the
    scene goes back to the traversal that replays the reference's order.  Forty over a bumpy grid with a light: batches,
    occlusion queries and the render against the checker, and the walk is what answered."""
    tris = _grid_mesh(16, lambda x, y: 0.08 * np.sin(8 * x) * np.cos(6 * y))
    body = " ".join("%r" % float(x) for x in tris.reshape(-1))
    rng = np.random.default_rng(23)
    centres = np.float32(rng.uniform((0.05, 0.05, 0.12), (0.95, 0.95, 0.6), (49, 3)))
    radii = np.float32(rng.uniform(0.02, 0.07, 49))

    def scene(n):
        balls = "".join("sphere {\n\tcenter [%r %r %r]\n\tradius %r\n}\n\n" % (*map(float, c), float(r)) for c, r in zip(centres[:n], radii[:n]))
        return ("render {\n\tcamera [0.5 0.5 2.4]\n\tlook_at [0.5 0.5 0]\n\tup [0 1 0]\n\twidth 56\n\theight 40\n\tsamples 12\n\tfov 30\n"
                "\tsky_box_color [.3 .4 .6]\n}\n\nmaterial {\n\tcolor [0.8 0.7 0.6]\n\troughness 0.3\n\tmetallic 1\n}\n\n" + balls +
                "material {\n\tcolor [0.7 0.8 0.6]\n\troughness 1\n}\n\ntriangle_mesh {\n\tvertices [ %s ]\n}\n\n"
                "material {\n\tcolor [0 0 0]\n\troughness 1\n}\n\narea_light {\n\tcolor [9 9 8]\n}\n\n"
                "triangle_mesh {\n\tvertices [ 0.4 0.4 1.0  0.6 0.4 1.0  0.5 0.6 1.05 ]\n}\n" % body)

    for n, walked in ((40, True), (49, False)):
        text = scene(n)
        osc = orc.Scene.parse(text).build_bvh()
        psc = rta.Scene.parse(text).build_bvh()
        gpu.upload(psc)
        nodes = osc.nodes()
        lo, hi = nodes["lo"][0], nodes["hi"][0]
        rr = raygen.random_rays(120000, lo - 0.2, hi + 0.2, seed=9 + n)
        sel = rng.random(rr.shape[0]) < 0.5      # half of them aimed at the spheres
        which = rng.integers(0, n, int(sel.sum()))
        target = centres[which] + rng.standard_normal((int(sel.sum()), 3)).astype(np.float32) * (radii[which, None] * np.float32(0.7))
        d = target - rr[sel, 0:3]
        rr[sel, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
        idx, t, u, v = osc.trace_closest(rr)
        ok = same_hits(batch(gpu, rr), dict(prim=idx, t=t, u=u, v=v))
        assert ok.all(), f"{n} spheres: {(~ok).sum()} rays differ"
        assert bool(gpu.last_batch_stats()["walked"]) == walked
        types = psc.arrays()["prim_type"]
        assert (types[idx[idx >= 0]] != 0).sum() > 5000
        k = rng.integers(0, 5, rr.shape[0])
        with np.errstate(over="ignore"):
            limit = np.where(idx >= 0, t * np.float32([0.6, 0.9999999, 1.0, 1.0000001, 1.7])[k], np.float32(3.0)).astype(np.float32)
        assert np.array_equal(gpu.trace_occluded(rr, limit), osc.trace_occluded(rr, limit))
        want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=8, samples_per_stream=4)
        bgr, acc, st = gpu.render_scene(psc, seed=8, samples_per_stream=4)
        assert st["rays"] == cn["rays"] and st["shadow_rays"] == cn["shadow_rays"] and (st["walked_rays"] > 0) == walked
        assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bgr, want_bgr)


def test_scenes_without_a_wide_tree_render_as_before(gpu):
    """Large triangles keep a scene on the exact traversal (cornell_box.txt's walls); variant 20 says so."""
    psc = rta.Scene.load(f"{orc.SCENES}/cornell_box.txt").build_bvh()
    gpu.upload(psc)
    with pytest.raises(rta.RtaError):
        walk(gpu, raygen.axis_rays())
    psc.set_render(64, 36, 4)
    _, _, st = gpu.render_scene(psc, seed=1, samples_per_stream=2)
    assert st["retraced_rays"] == 0 and st["walked_rays"] == 0


def test_shadow_rays_take_the_any_hit_walk(gpu):
    """A triangle-only scene with a light: next-event estimation queues occlusion queries, which the any-hit walk
    answers (scenes with a wide tree); sums and image bytes are the checker's, with and without the wide tree."""
    tris = _grid_mesh(16, lambda x, y: 0.15 * np.sin(6 * x) * np.cos(4 * y))
    body = " ".join("%r" % float(x) for x in tris.reshape(-1))
    text = ("render {\n\tcamera [0.5 0.5 2.5]\n\tlook_at [0.5 0.5 0]\n\tup [0 1 0]\n\twidth 64\n\theight 48\n\tsamples 32\n\tfov 25\n"
            "\tsky_box_color [.05 .06 .1]\n}\n\nmaterial {\n\tcolor [0.8 0.7 0.6]\n\troughness 1\n}\n\n"
            "triangle_mesh {\n\tvertices [ %s ]\n}\n\n"
            "material {\n\tcolor [0 0 0]\n\troughness 1\n}\n\narea_light {\n\tcolor [12 11 9]\n}\n\n"
            "triangle_mesh {\n\tvertices [ 0.3 0.3 1.2  0.7 0.3 1.2  0.5 0.7 1.3   0.1 0.9 0.9  0.3 0.9 0.9  0.2 0.7 1.0 ]\n}\n" % body)
    osc = orc.Scene.parse(text).build_bvh()
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=5, samples_per_stream=4)
    for flags in (0, rta.RENDER_EXACT_ONLY):
        bgr, acc, st = gpu.render_scene(psc, seed=5, samples_per_stream=4, flags=flags)
        assert st["rays"] == cn["rays"] and st["shadow_rays"] == cn["shadow_rays"] and st["shadow_rays"] > 10000
        assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bgr, want_bgr)
        assert (st["walked_rays"] > 0) == (flags == 0)


def test_large_triangles_get_no_wide_tree(gpu):
    """The walk's padding grows with the square of the triangles' edges: a room of twelve big triangles keeps the
    replaying traversal, and renders as the checker does."""
    c = [(x, y, z) for x in (0.0, 2.0) for y in (0.0, 2.0) for z in (0.0, 2.0)]
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = np.float32([c[q[0]] + c[q[1]] + c[q[2]] for q in quads] + [c[q[0]] + c[q[2]] + c[q[3]] for q in quads])
    text = _mesh_scene(tris).replace("camera [0.5 0.5 3]", "camera [1 1 1]").replace("look_at [0.5 0.5 0]", "look_at [1.9 1.2 0.3]")
    osc = orc.Scene.parse(text).build_bvh()
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=2, samples_per_stream=4)
    bgr, acc, st = gpu.render_scene(psc, seed=2, samples_per_stream=4)
    assert st["walked_rays"] == 0 and st["rays"] == cn["rays"] > 2 * cn["samples"]
    assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bgr, want_bgr)


def test_rays_that_travel_together_walk_the_tree_together(gpu, monkeypatch):
    """A batch in order whose neighbouring rays travel together (a camera's pixels) takes the wide tree 64 rays at
    a time (csrc/traverse_packet.hpp): same hits as the checker's, bit for bit, and the statistics say which walk
    ran.  Forced onto rays that do NOT travel together - random ones, and the dragon's, whose tree is too small
    for the choice to be made at all - the packets are slow but still exact."""
    import torch
    from raytracer_amd.synthetic import make_soup
    tris = make_soup(300_000, seed=13)
    psc = rta.Scene.from_triangles(tris).build_bvh()
    osc = orc.Scene.from_triangles(tris).build_bvh()
    gpu.upload(psc)
    dev = torch.device("cuda", 0)

    def traced(rr):
        rt = torch.from_numpy(np.ascontiguousarray(rr, np.float32)).to(dev)
        ht = torch.zeros((rr.shape[0], 4), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        gpu.trace_closest_device(rt.data_ptr(), rr.shape[0], ht.data_ptr())
        return ht.cpu().numpy(), gpu.last_batch_stats()

    def checked(rr, got):
        idx, t, u, v = osc.trace_closest(rr)
        assert same_hits(got, dict(prim=idx, t=t, u=u, v=v)).all()
        return idx

    # a pinhole grid from outside, 512 x 512, and the same grid from inside the soup
    side = 512
    t = (np.arange(side, dtype=np.float32) + 0.5) / side * 2 - 1
    y, z = np.meshgrid(t * 0.05, -t * 0.05)       # 64 neighbouring rays within 0.7 degrees
    d = np.stack([np.ones_like(y), y, z], axis=-1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for origin in ((-3.0, 0.0, 0.0), (0.1, 0.2, -0.3)):
        rr = np.concatenate([np.broadcast_to(np.float32(origin), d.shape), d], axis=1).astype(np.float32)
        got, st = traced(rr)
        assert st["walked"] == 1 and st["packets"] == 1 and st["sorted"] == 1, st
        idx = checked(rr, got)
        assert (idx >= 0).mean() > 0.2
    # rays in order that do not travel together, or that start in different places, keep the walk of their own
    rng = np.random.default_rng(5)
    spread = rr.copy()
    spread[:, 3:6] = rng.standard_normal((len(rr), 3)).astype(np.float32)
    parallel = rr.copy()
    parallel[:, 0:3] = np.stack([np.full(len(rr), -3.0), y.reshape(-1) * 10, z.reshape(-1) * 10], axis=1)
    parallel[:, 3:6] = np.float32([1, 0, 0])
    for other in (spread, parallel):
        got, st = traced(other)
        assert st["packets"] == 0, st
        checked(other, got)
    # forced: random rays in packets
    monkeypatch.setenv("RTA_BATCH_PACKETS", "2")
    lo, hi = np.full(3, -1.1), np.full(3, 1.1)
    rr = raygen.random_rays(300_000, lo, hi, seed=77).astype(np.float32)
    got, st = traced(rr)
    assert st["packets"] == 1
    checked(rr, got)
    # ... and on the dragon: rays inside it, rays outside the walk's reach, zero components, NaN
    dsc = orc.Scene.load(f"{orc.SCENES}/dragon.txt").build_bvh()
    gpu.upload(rta.Scene.load(f"{orc.SCENES}/dragon.txt").build_bvh())
    nodes = dsc.nodes()
    rr = raygen.random_rays(400_000, nodes["lo"][0] - 0.3, nodes["hi"][0] + 0.3, seed=78).astype(np.float32)
    rr[::97, 0:3] *= 40.0
    rr[::101, 4] = 0.0
    rr[::1009, 3] = np.nan
    got, st = traced(rr)
    assert st["packets"] == 1
    idx, t, u, v = dsc.trace_closest(rr)
    assert same_hits(got, dict(prim=idx, t=t, u=u, v=v)).all()


def test_a_mesh_on_a_floor_of_large_triangles_takes_the_walk(gpu):
    """dragon.txt standing on two floor triangles eight units wide.  The reference's midpoint splits leave half the
    mesh in ONE leaf beside a floor triangle (its box spans the scene, every split puts all centroids on one
    side), which the order-replaying traversal has to plough through; the wide tree has a topology of its own, so
    the walk answers nearly every ray and only what it hands on meets that leaf.  Same bits as the checker."""
    text = open(f"{orc.SCENES}/dragon.txt").read().replace("width 400", "width 64").replace("height 400", "height 40").replace("samples 25", "samples 8")
    text += ("\nmaterial {\n\tcolor [0.6 0.6 0.6]\n\troughness 1\n}\n\ntriangle_mesh {\n\tvertices [ -4 -4 0.27  4 -4 0.27  4 4 0.27   "
             "-4 -4 0.27  4 4 0.27  -4 4 0.27 ]\n}\n")
    osc = orc.Scene.parse(text).build_bvh()
    psc = rta.Scene.parse(text).build_bvh()
    assert psc.arrays()["node_count"].max() > 1000          # the long leaf
    gpu.upload(psc)
    want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=3, samples_per_stream=4)
    bgr, acc, st = gpu.render_scene(psc, seed=3, samples_per_stream=4)
    assert st["rays"] == cn["rays"] and st["walked_rays"] > 0.9 * st["rays"] and st["retraced_rays"] < 0.05 * st["rays"]
    assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bgr, want_bgr)
