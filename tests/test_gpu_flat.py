"""Scenes of a few dozen primitives, whose rays test every primitive instead of walking a tree
(csrc/traverse_flat.hpp: k_intersect_flat, k_shadow_flat, k_batch_flat, k_batch_occluded_flat): answers against the
CPU checker's, bit for bit, on rays chosen to sit where the division-free decisions are closest to the reference's
divisions - shared edges and vertices (u = 0, v = 0, u + v = 1), exact ties between coplanar copies, origins at the
1e-4 cut of the distance, limits at the very distance of a hit - and on whole renders with lights.  The reference's
test: src/main.cpp:250-279; its traversals: src/main.cpp:332-422, 424-478."""
import numpy as np
import pytest

import orc
import raytracer_amd as rta

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _vec(v):
    return "[" + " ".join("%r" % float(x) for x in v) + "]"


_HEAD = ("render {\n\tcamera [-2.6 0.3 0.9]\n\tlook_at [0 0 0.6]\n\twidth 48\n\theight 40\n\tsamples 6\n\tsky_box_color [.4 .5 .8]\n}\n\n"
         "material {\n\tcolor [0.7 0.7 0.6]\n\troughness 0.8\n}\n\n")


def _mesh(tris):
    body = " ".join("%r" % float(x) for x in np.float32(tris).reshape(-1))
    return f"triangle_mesh {{\n\tvertices [ {body} ]\n}}\n\n"


def _small_scene(seed, n_tris=20, n_spheres=3, light=True, copies=False):
    """A floor of two large triangles, a fan of triangles around a shared vertex (shared edges), random triangles,
    spheres, a light; `copies`: two of the triangles twice (exact ties)."""
    rng = np.random.default_rng(seed)
    floor = [(-1.5, -1.5, 0, 1.5, -1.5, 0, 1.5, 1.5, 0), (-1.5, -1.5, 0, 1.5, 1.5, 0, -1.5, 1.5, 0)]
    hub = (0.1, 0.05, 0.9)
    rim = [(0.1 + 0.7 * np.cos(a), 0.05 + 0.7 * np.sin(a), 0.5 + 0.1 * np.sin(3 * a)) for a in np.linspace(0, 2 * np.pi, 7)[:-1]]
    fan = [hub + rim[k] + rim[(k + 1) % 6] for k in range(6)]
    loose = []
    for _ in range(max(n_tris - 8, 0)):
        c = rng.uniform(-1, 1, 3) * (1.0, 1.0, 0.5) + (0, 0, 0.8)
        loose.append(tuple(np.concatenate([c + rng.uniform(-0.4, 0.4, 3) for _ in range(3)])))
    tris = floor + fan + loose
    if copies:
        tris = tris + [fan[2], loose[0]]
    text = _HEAD + _mesh(tris)
    text += "material {\n\tcolor [0.8 0.3 0.2]\n\troughness 0.2\n\tmetallic 1\n}\n\n"
    for _ in range(n_spheres):
        c = rng.uniform(-0.9, 0.9, 3) * (1, 1, 0.4) + (0, 0, 0.5)
        text += f"sphere {{\n\tcenter {_vec(c)}\n\tradius {rng.uniform(0.1, 0.3)!r}\n}}\n\n"
    if light:
        text += ("material {\n\tcolor [0 0 0]\n}\n\narea_light {\n\tcolor [9 8 6]\n}\n\n" +
                 _mesh([(-0.3, -0.3, 1.9, 0.3, -0.3, 1.9, 0.3, 0.3, 1.9), (-0.3, -0.3, 1.9, 0.3, 0.3, 1.9, -0.3, 0.3, 1.9)]))
    return text


def _pair(text):
    return orc.Scene.parse(text).build_bvh(), rta.Scene.parse(text).build_bvh()


def _aimed_rays(osc, n, seed):
    """Rays from random origins at points ON the scene's triangles: vertices, points of edges, and interior points, and
    (a third of them) started on a triangle's plane at 1e-4 and a few units in the last place around it."""
    rng = np.random.default_rng(seed)
    objs = osc.objects()
    tris = np.float32(objs["geom"][objs["type"] == 0][:, :9]).reshape(-1, 3, 3)      # v0, v1, v2
    which = rng.integers(0, len(tris), n)
    kind = rng.integers(0, 4, n)
    w = rng.random((n, 3)).astype(np.float32)
    w[kind == 0] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, (kind == 0).sum())]       # a vertex
    edge = kind == 1
    w[edge, rng.integers(0, 3, edge.sum())] = 0                                           # a point of an edge
    w /= w.sum(axis=1, keepdims=True)
    target = np.einsum("nk,nkc->nc", w, tris[which]).astype(np.float32)
    origin = (rng.uniform(-1.4, 1.4, (n, 3)) * (1, 1, 0.6) + (0, 0, 1.0)).astype(np.float32)
    d = target - origin
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    near = kind == 3
    # origins a hair's breadth in front of the target: the distance lands on either side of the reference's 1e-4
    step = np.float32(1e-4) * (1 + rng.integers(-6, 7, n).astype(np.float32) * np.float32(2.0 ** -23))
    origin[near] = (target[near] - d[near] * step[near, None]).astype(np.float32)
    return np.concatenate([origin, d], axis=1).astype(np.float32)


def _check_batches(gpu, text, seed, expect_retraced=None):
    osc, psc = _pair(text)
    gpu.upload(psc)
    nodes = osc.nodes()
    lo, hi = nodes["lo"][0], nodes["hi"][0]
    rng = np.random.default_rng(seed)
    n = 300_000
    o = rng.uniform(lo - 0.5, hi + 0.5, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rr = np.concatenate([np.concatenate([o, d], axis=1), _aimed_rays(osc, n, seed + 1)]).astype(np.float32)
    # a few rays the scan must hand on: zero components, a huge origin
    rr[:4, 3:] = np.float32([[1, 0, 0], [0, 1, 0], [0, 0, -1], [0.6, 0.8, 0]])
    rr[4, :3] = np.float32([3e12, 0, 0.5])
    idx, t, u, v = osc.trace_closest(rr)
    hits = gpu.trace_closest(rr)
    st = gpu.last_batch_stats()
    assert st["scanned"] == 1 and st["walked"] == 0, st
    assert np.array_equal(hits["prim"], idx), int((hits["prim"] != idx).sum())
    hit = idx >= 0
    assert hit.mean() > 0.3
    assert np.array_equal(bits(hits["t"]), bits(t))
    assert np.array_equal(bits(hits["u"][hit]), bits(u[hit])) and np.array_equal(bits(hits["v"][hit]), bits(v[hit]))
    if expect_retraced is not None:
        assert expect_retraced(st["retraced_rays"], rr.shape[0]), st
    # occlusion: random limits, and limits at the very distance of the closest hit and its neighbours
    lim = (rng.random(rr.shape[0], dtype=np.float32) * np.float32(3.0)).astype(np.float32)
    at = hit & (rng.random(rr.shape[0]) < 0.5)
    ulps = rng.integers(-3, 4, rr.shape[0])
    lim[at] = (t[at].view(np.int32) + ulps[at]).astype(np.int32).view(np.float32)
    want = osc.trace_occluded(rr, lim)
    got = gpu.trace_occluded(rr, lim)
    assert np.array_equal(got, want), int((got != want).sum())
    assert 0.1 < want.mean() < 0.9
    return st


def test_rays_at_edges_vertices_and_the_distance_cut(gpu):
    """600 k rays per scene, half of them aimed at vertices, edges and the 1e-4 cut: every answer the checker's.  The
    scan hands on what it is built to hand on (rays through a shared vertex or edge meet two triangles at distances it
    cannot tell apart, rays at the cut a distance it cannot place) and answers the rest itself."""
    for seed in (1, 2, 3):
        st = _check_batches(gpu, _small_scene(seed), seed, expect_retraced=lambda r, n: n // 100 < r < n // 4)
    # spheres only / triangles only / the largest scene that is scanned
    _check_batches(gpu, _small_scene(4, n_tris=8, n_spheres=9, light=False), 4)
    _check_batches(gpu, _small_scene(5, n_tris=61, n_spheres=0, light=True), 5)


def test_exact_ties_go_to_the_replaying_traversal(gpu):
    """Two triangles of the scene twice: a ray that hits one hits its copy at the same distance, and which of the two the
    reference reports depends on its visit order - such rays are handed on, and still come out as the checker has them."""
    st = _check_batches(gpu, _small_scene(6, copies=True), 6, expect_retraced=lambda r, n: r > n // 200)


def test_a_scene_too_large_to_scan_takes_a_tree(gpu):
    osc, psc = _pair(_small_scene(7, n_tris=80, n_spheres=0, light=False))
    gpu.upload(psc)
    rr = _aimed_rays(osc, 50_000, 8)
    idx, t, u, v = osc.trace_closest(rr)
    hits = gpu.trace_closest(rr)
    assert gpu.last_batch_stats()["scanned"] == 0
    assert np.array_equal(hits["prim"], idx) and np.array_equal(bits(hits["t"]), bits(t))


@pytest.mark.parametrize("seed,copies", [(11, False), (12, True)])
def test_renders_with_lights_and_spheres(gpu, seed, copies):
    """Whole renders (bounce rays and occlusion queries through the scan, camera rays from their cells' lists or, for a
    lens camera, through the scan as well) against the checker's, and the same frame with the scan switched off."""
    text = _small_scene(seed, copies=copies)
    for lens in ("", "\tdefocus_angle 0.6\n"):
        t = text.replace("\tsamples 6\n", "\tsamples 6\n" + lens)
        osc, psc = _pair(t)
        gpu.upload(psc)
        want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=seed, samples_per_stream=3)
        bgr, acc, st = gpu.render_scene(psc, seed=seed, samples_per_stream=3)
        assert st["rays"] == cn["rays"] and st["samples"] == cn["samples"]
        assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bgr, want_bgr)
        assert st["scanned_rays"] > 0 and st["walked_rays"] == 0 and st["shadow_rays"] > 0, st
        assert (st["listed_rays"] == 0) == bool(lens)
        assert st["retraced_rays"] < st["rays"] // 10 if not copies else st["retraced_rays"] > 0
        bgr2, acc2, st2 = gpu.render_scene(psc, seed=seed, samples_per_stream=3, flags=rta.RENDER_EXACT_ONLY)
        assert st2["scanned_rays"] == 0 and np.array_equal(bits(acc2), bits(acc))


def test_a_primitive_that_no_leaf_holds_is_never_reported(gpu):
    """A description may leave primitives out of every leaf; the reference never tests those.  The scan tests every
    record, so a ray aimed at such a triangle must come back as the replaying traversal has it (which is compared with
    the checker everywhere else): from behind the triangle, or not at all."""
    import torch
    osc, psc = _pair(_small_scene(9, n_tris=24, n_spheres=1, light=False))
    a = psc.arrays()
    leaves = np.nonzero(a["node_count"] > 1)[0]
    assert leaves.size > 0
    node = int(leaves[0])
    dropped = int(a["node_index"][node] + a["node_count"][node] - 1)      # the leaf's last primitive, in the scene's order
    a["node_count"] = a["node_count"].copy()
    a["node_count"][node] -= 1
    desc = rta.desc_from_arrays(a)
    gpu.upload(desc)
    tri = np.float32(a["prim_geom"]).reshape(-1, 9)[dropped].reshape(3, 3)
    assert a["prim_type"][dropped] == 0
    rng = np.random.default_rng(3)
    n = 100_000
    w = rng.random((n, 3)).astype(np.float32)
    w /= w.sum(axis=1, keepdims=True)
    target = (w @ tri).astype(np.float32)
    origin = (rng.uniform(-1.4, 1.4, (n, 3)) * (1, 1, 0.6) + (0, 0, 1.0)).astype(np.float32)
    d = target - origin
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    rr = np.concatenate([origin, d], axis=1).astype(np.float32)
    dev = torch.device("cuda", 0)
    rt = torch.from_numpy(rr).to(dev)
    got = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    want = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu.trace_closest_device(rt.data_ptr(), n, got.data_ptr())
    assert gpu.last_batch_stats()["scanned"] == 1
    gpu.trace_closest_device(rt.data_ptr(), n, want.data_ptr(), exact_only=True)
    assert gpu.last_batch_stats()["scanned"] == 0
    g, e = got.cpu().numpy(), want.cpu().numpy()
    assert np.array_equal(g.view(np.uint32), e.view(np.uint32))
    assert not (e[:, 0].view(np.int32) == dropped).any()
