"""The division-free triangle classification (csrc/traverse.hpp: classify_triangle) restated in numpy binary32 and
held against the reference's test (src/main.cpp:250-279), which is restated the same way and first checked against
the CPU checker's closest hits: whatever the classification rejects the reference rejects, whatever it accepts the
reference accepts at a distance within 2^-21 of the approximation, and what it leaves open is rare.  The hardware's
reciprocal is good to one unit in the last place, so the approximation is tried with the correctly rounded
reciprocal and with both of its neighbours.  No GPU needed: this is the arithmetic behind the margins."""
import numpy as np
import pytest

import orc

F = np.float32
FLT_MAX = np.finfo(np.float32).max


def _cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)


def _dot(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]


def _numerators(v0, v1, v2, o, d):
    """det and the numerators of u, v, t in the reference's order of operations, all binary32."""
    e0, e1 = v0 - v2, v1 - v2
    pvec = _cross(d, e1)
    det = _dot(e0, pvec)
    tvec = o - v2
    a = _dot(tvec, pvec)
    qvec = _cross(tvec, e0)
    b = _dot(d, qvec)
    c = _dot(e1, qvec)
    return det, a, b, c


def hit_triangle(v0, v1, v2, o, d):
    """src/main.cpp:250-279 for arrays of rays and triangles: t (FLT_MAX for a miss)."""
    det, a, b, c = _numerators(v0, v1, v2, o, d)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        u, v, t = a / det, b / det, c / det
        miss = (det > F(-.0001)) & (det < F(.0001))
        miss |= (u < 0) | (u > 1)
        miss |= (v < 0) | (u + v > 1)
        miss |= ~(t > F(.0001))
    return np.where(miss, FLT_MAX, t).astype(F)


def classify(v0, v1, v2, o, d, ulps=0):
    """classify_triangle: (rejected, accepted, q); `ulps`: the reciprocal's distance from the correctly rounded one."""
    det, a, b, c = _numerators(v0, v1, v2, o, d)
    sign = det.view(np.uint32) & np.uint32(0x80000000)
    D = np.abs(det)
    A = (a.view(np.uint32) ^ sign).view(F)
    B = (b.view(np.uint32) ^ sign).view(F)
    C = (c.view(np.uint32) ^ sign).view(F)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        s = A + B
        r = (F(1) / D).astype(F)
        r = (r.view(np.int32) + np.int32(ulps)).view(F)
        q = C * r
        tiny = D * F(-2.0 ** -100)
        rejected = ~(D >= F(.0001)) | (A < tiny) | (A > D) | (B < tiny) | (s > D * F(1 + 2.0 ** -20)) | ~(q >= F(.0001) * F(1 - 2.0 ** -20))
        accepted = ~rejected & (A >= 0) & (B >= 0) & (s < D * F(1 - 2.0 ** -20)) & (q > F(.0001) * F(1 + 2.0 ** -20))
    return rejected, accepted, q


def _unit(d):
    return (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(F)


def _pairs(n, seed, scale=1.0):
    """Triangles of mixed sizes with rays at them: a quarter anywhere, a quarter at vertices, a quarter at points of edges,
    a quarter from 1e-4 and a few units in the last place in front of a point of the triangle."""
    rng = np.random.default_rng(seed)
    size = (10.0 ** rng.uniform(-2.5, 0.3, (n, 1))).astype(F)
    c = rng.uniform(-1, 1, (n, 3)).astype(F)
    v = [(c + rng.uniform(-1, 1, (n, 3)).astype(F) * size).astype(F) for _ in range(3)]
    tri = np.stack(v, axis=1)
    kind = rng.integers(0, 4, n)
    w = rng.random((n, 3)).astype(F)
    w[kind == 1] = np.eye(3, dtype=F)[rng.integers(0, 3, (kind == 1).sum())]
    edge = kind == 2
    w[edge, rng.integers(0, 3, edge.sum())] = 0
    w /= w.sum(axis=1, keepdims=True)
    target = np.einsum("nk,nkc->nc", w, tri).astype(F)
    target[kind == 0] += (rng.uniform(-1.5, 1.5, ((kind == 0).sum(), 3)) * size[kind == 0]).astype(F)
    o = rng.uniform(-1.5, 1.5, (n, 3)).astype(F)
    d = _unit(target - o)
    near = kind == 3
    step = F(1e-4) * (1 + rng.integers(-6, 7, n).astype(F) * F(2.0 ** -23))
    o[near] = (target[near] - d[near] * step[near, None]).astype(F)
    s = F(scale)
    return (tri[:, 0] * s).astype(F), (tri[:, 1] * s).astype(F), (tri[:, 2] * s).astype(F), (o * s).astype(F), d, kind


def test_the_restated_reference_test_is_the_checkers():
    """One-triangle scenes (the root is a leaf: no box is tested) through the CPU checker's closest hit."""
    rng = np.random.default_rng(5)
    for k in range(6):
        v0, v1, v2, o, d, _ = _pairs(40_000, 100 + k)
        # (a triangle large enough to be seen at all: the reference's determinant cut is absolute)
        tri = rng.uniform(-1, 1, 9).astype(F) * F(0.3 * 3.0 ** k)
        body = " ".join("%r" % float(x) for x in tri)
        text = ("render {\n\tcamera [-3 0 1]\n\tlook_at [0 0 1]\n\twidth 8\n\theight 8\n\tsamples 1\n}\n\nmaterial {\n\tcolor [1 1 1]\n}\n\n"
                f"triangle_mesh {{\n\tvertices [ {body} ]\n}}\n")
        osc = orc.Scene.parse(text).build_bvh()
        # rays made for other triangles miss this one mostly: aim half of them at it
        n = o.shape[0]
        w = rng.random((n, 3)).astype(F)
        w /= w.sum(axis=1, keepdims=True)
        target = (w @ tri.reshape(3, 3)).astype(F)
        d[: n // 2] = _unit(target[: n // 2] - o[: n // 2])
        rr = np.concatenate([o, d], axis=1).astype(F)
        idx, t, u, v = osc.trace_closest(rr)
        one = np.broadcast_to(tri.reshape(3, 3), (n, 3, 3))
        want = hit_triangle(one[:, 0], one[:, 1], one[:, 2], o, d)
        assert np.array_equal(np.where(idx >= 0, t, FLT_MAX).astype(F).view(np.uint32), want.view(np.uint32))
        assert (idx >= 0).mean() > 0.2


@pytest.mark.parametrize("scale", [1.0, 0.03, 40.0])
def test_verdicts_are_the_references(scale):
    """4 M (triangle, ray) pairs per scale, three quarters of them aimed at what the margins are about: rejected means the
    reference misses, accepted means it hits at a distance the approximation has to 2^-21, whichever way the hardware's
    reciprocal errs.  Left open: the pairs built to be (rays at the very distance cut, most rays aimed at an edge) and, of
    the rays aimed anywhere, fewer than one in ten thousand."""
    n_open = n_all = n_open_anywhere = n_anywhere = 0
    for seed in range(4):
        v0, v1, v2, o, d, kind = _pairs(1_000_000, 1000 * seed + int(scale * 10), scale)
        t = hit_triangle(v0, v1, v2, o, d)
        hit = t < FLT_MAX
        assert 0.03 < hit.mean() < 0.9      # (at the small scale most triangles fall to the absolute determinant cut)
        for ulps in (-1, 0, 1):
            rejected, accepted, q = classify(v0, v1, v2, o, d, ulps)
            assert not (rejected & hit).any(), int((rejected & hit).sum())
            assert not (accepted & ~hit).any(), int((accepted & ~hit).sum())
            rel = np.abs(q[accepted].astype(np.float64) / t[accepted].astype(np.float64) - 1)
            assert rel.max() <= 2.0 ** -21, rel.max()
            n_open += int((~rejected & ~accepted).sum())
            n_all += rejected.size
            n_open_anywhere += int((~rejected & ~accepted & (kind == 0)).sum())
            n_anywhere += int((kind == 0).sum())
    assert n_open < n_all * 0.35, (n_open, n_all)
    assert n_open_anywhere < n_anywhere * 1e-4, (n_open_anywhere, n_anywhere)
