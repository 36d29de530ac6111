"""Rays aimed at the regime the certified walk's padding is derived for
(csrc/wide_tree.cpp): determinants just above and just below the reference's
absolute cut of 1e-4 (src/main.cpp:250-279), targets on edges and vertices,
origins on the sphere of the farthest admitted origin, directions of length
0.5, 1 and just under 2 (the longest the walk admits), and the same mesh
re-scaled so that E^2 (S + E) - the quantity
the padding grows with - moves by orders of magnitude.

Two statements per batch, both against the CPU checker, bit for bit:
  * what rta_trace_closest returns (the walk, then the exact traversal for the
    rays the walk leaves in doubt) is the checker's answer for every ray;
  * the walk ALONE (probe variant 21) never returns a wrong answer: a ray is
    either handed on (prim = -2) or answered exactly as the checker answers it.
    A certified answer that differs would be a hole in the proof, which the
    exact traversal behind it could not mend."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

import orc
import raytracer_amd as rta
from raytracer_amd.synthetic import make_soup

pytestmark = pytest.mark.gpu


def aimed_rays(tris, centre, reach, n, seed, dir_len):
    """n rays against randomly chosen triangles of `tris` ((m, 9) float32)."""
    rng = np.random.default_rng(seed)
    t = tris[rng.integers(0, len(tris), n)].astype(np.float64).reshape(n, 3, 3)
    e0, e1 = t[:, 0] - t[:, 2], t[:, 1] - t[:, 2]
    nrm = np.cross(e0, e1)
    area2 = np.linalg.norm(nrm, axis=1)
    ok = area2 > 0
    nrm[ok] /= area2[ok, None]
    # the target: inside, on an edge, on a vertex
    b = rng.dirichlet((1, 1, 1), n)
    mode = rng.integers(0, 3, n)
    edge = mode == 1
    b[edge, rng.integers(0, 3, edge.sum())] = 0
    b[edge] /= b[edge].sum(axis=1, keepdims=True)
    vertex = mode == 2
    b[vertex] = np.eye(3)[rng.integers(0, 3, vertex.sum())]
    p = (t * b[:, :, None]).sum(axis=1)
    # half of the targets moved by a few units in the last place, to either side of the edge
    nudge = rng.random(n) < 0.5
    p[nudge] += rng.standard_normal((nudge.sum(), 3)) * 2e-7 * (np.abs(p[nudge]).max(axis=1, keepdims=True) + 1e-3)
    # the angle to the normal from the determinant wanted: |det| = |d| cos(angle) |e0 x e1|, from half the cut to
    # four times the cut; where the triangle is too small for that, anything down to grazing
    det = 10.0 ** rng.uniform(np.log10(0.5e-4), np.log10(4e-4), n)
    cos = det / (dir_len * np.maximum(area2, 1e-30))
    cos = np.where(cos <= 1, cos, rng.uniform(0.02, 1.0, n))
    sin = np.sqrt(np.maximum(0.0, 1 - cos * cos))
    w = np.cross(nrm, rng.standard_normal((n, 3)))
    w /= np.maximum(np.linalg.norm(w, axis=1, keepdims=True), 1e-30)
    d = (nrm * cos[:, None] + w * sin[:, None]) * rng.choice([-1.0, 1.0], (n, 1))
    # half of the origins on the sphere of the farthest admitted origin, half nearby
    q = p - centre
    qd = (q * d).sum(axis=1)
    radius = reach * (1 - 1e-4)
    far = qd + np.sqrt(np.maximum(qd * qd - (q * q).sum(axis=1) + radius * radius, 0.0))
    near = rng.uniform(0.02, 0.6, n) * reach
    s = np.where(rng.random(n) < 0.5, far, np.minimum(near, far))
    o = p - d * s[:, None]
    return np.concatenate([o, d * dir_len], axis=1).astype(np.float32)


def run_variant(gpu, variant, rr):
    dev = torch.device("cuda", 0)
    rt = torch.from_numpy(rr).to(dev)
    ht = torch.zeros((rr.shape[0], 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu.trace_variant_device(variant, rt.data_ptr(), rr.shape[0], ht.data_ptr())
    return ht.cpu().numpy()


def checker_hits(osc, rr, pool):
    pieces = np.array_split(np.arange(len(rr)), 16)
    want = list(pool.map(lambda ix: osc.trace_closest(rr[ix]), pieces))
    return [np.concatenate([x[k] for x in want]) for k in range(4)]


def same(got, want):
    idx, t, u, v = want
    ok = got[:, 0].view(np.int32) == idx
    ok &= got[:, 1].view(np.uint32) == t.view(np.uint32)
    hit = idx >= 0
    ok &= ~hit | ((got[:, 2].view(np.uint32) == u.view(np.uint32)) & (got[:, 3].view(np.uint32) == v.view(np.uint32)))
    return ok


def dragon_triangles():
    sc = rta.Scene.load(f"{orc.SCENES}/dragon.txt").build_bvh()
    return sc.arrays()["prim_geom"].copy()


CASES = [
    # name, scale, rays per direction length, a wide tree is expected
    ("dragon", 1.0, 3_400_000, True),
    ("dragon x 0.5", 0.5, 700_000, True),
    ("dragon x 2", 2.0, 700_000, True),
    ("dragon x 100", 100.0, 700_000, False),     # edges of 6: no padding the walk could live with
    ("dragon x 0.01", 0.01, 700_000, True),      # every determinant far below the cut: nothing is ever hit
    ("soup", 1.0, 700_000, True),
]


@pytest.mark.parametrize("name,scale,n,wide", CASES, ids=[c[0] for c in CASES])
def test_rays_aimed_at_the_padding_bound(gpu, name, scale, n, wide):
    tris = make_soup(300_000, seed=11) if name == "soup" else (dragon_triangles() * np.float32(scale)).astype(np.float32)
    scene = rta.Scene.from_triangles(tris).build_bvh()
    osc = orc.Scene.from_triangles(tris).build_bvh()
    gpu.upload(scene)
    geom = scene.arrays()["prim_geom"]      # in tree order, as the hits number them
    scalars = gpu.scene_buffer(6).view(np.uint32)
    assert bool(scalars[11]) == wide, "wide tree expected" if wide else "no wide tree expected"
    nodes = osc.nodes()
    lo, hi = nodes["lo"][0].astype(np.float64), nodes["hi"][0].astype(np.float64)
    centre = 0.5 * (lo + hi)
    reach = float(scalars[14:15].view(np.float32)[0]) if wide else 2.0 * float(np.linalg.norm(hi - lo))
    handed_on = hits = 0
    with ThreadPoolExecutor(16) as pool:
        for k, dir_len in enumerate((0.5, 1.0, 1.9999)):      # (the walk admits |d|^2 <= 4 (1 - 2^-20))
            rr = aimed_rays(geom, centre, reach, n, seed=100 + k, dir_len=dir_len)
            want = checker_hits(osc, rr, pool)
            final = run_variant(gpu, 0, rr)
            ok = same(final, want)
            assert ok.all(), f"{name}, |d| = {dir_len}: {(~ok).sum()} rays differ from the checker, first {rr[np.nonzero(~ok)[0][0]].tolist()}"
            hits += int((want[0] >= 0).sum())
            if wide:
                alone = run_variant(gpu, 21, rr)
                doubt = alone[:, 0].view(np.int32) == -2
                ok = same(alone, want) | doubt
                assert ok.all(), (f"{name}, |d| = {dir_len}: the walk certified {(~ok).sum()} wrong answers, first "
                                  f"{rr[np.nonzero(~ok)[0][0]].tolist()}")
                handed_on += int(doubt.sum())
    print(f"{name}: {3 * n} rays, {hits} hits, {handed_on} handed on by the walk")
    if name in ("dragon", "soup"):
        assert hits > 0.05 * 3 * n          # the batch does reach the triangles it aims at
    if name == "dragon x 0.01":
        assert hits == 0
    if wide:
        assert handed_on < 0.05 * 3 * n     # the walk, not the traversal behind it, answered these batches
