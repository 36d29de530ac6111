"""rta_upload_scene's two paths: layouts and the wide tree made on the host
(small scenes) or by kernels on the GPU (scenes beyond 2 Mi primitives;
RTA_UPLOAD_DEVICE forces either).  Both must leave the same bytes on the
device - the device path is the host path's arithmetic, restated as kernels -
and what the device path does not handle must come out of the host path as
before.  The 10 M-triangle test in test_gpu_full_size.py takes the device path
by default and checks its hits against the CPU checker."""
import numpy as np
import pytest

import orc
import rays as raygen
import raytracer_amd as rta
from raytracer_amd.synthetic import make_soup

pytestmark = pytest.mark.gpu

NAMES = ["pair records", "primitive records", "shading records", "reference nodes", "wide nodes", "leaf boxes", "scalars"]


def layouts(gpu):
    return [gpu.scene_buffer(k) for k in range(7)]


def both_paths(gpu, scene, monkeypatch):
    # the host path at these sizes gives the wide tree a topology of its own; with RTA_WIDE_MIDPOINT
    # it keeps the reference's, which is what the device path builds on
    monkeypatch.setenv("RTA_WIDE_MIDPOINT", "1")
    monkeypatch.setenv("RTA_UPLOAD_DEVICE", "0")
    gpu.upload(scene)
    host = layouts(gpu)
    monkeypatch.setenv("RTA_UPLOAD_DEVICE", "1")
    gpu.upload(scene)
    return host, layouts(gpu)


def grid_mesh(n):
    """A bumpy n x n grid of quads whose vertices lie on a binary grid: exact ties, flat boxes."""
    rng = np.random.default_rng(3)
    z = np.round(rng.random((n + 1, n + 1)) * 4) / 64
    tris = []
    for i in range(n):
        for j in range(n):
            p = lambda a, b: (a / 8.0, b / 8.0, z[a, b])
            tris.append(p(i, j) + p(i + 1, j) + p(i + 1, j + 1))
            tris.append(p(i, j) + p(i + 1, j + 1) + p(i, j + 1))
    return np.asarray(tris, np.float32)


@pytest.mark.parametrize("which", ["dragon", "soup300k", "grid", "coincident"])
def test_device_upload_leaves_the_host_paths_bytes(gpu, which, monkeypatch):
    tris = None
    if which == "dragon":
        scene = rta.Scene.load(f"{orc.SCENES}/dragon.txt").build_bvh()
    elif which == "soup300k":
        tris = make_soup(300_000, seed=5)
    elif which == "grid":
        tris = grid_mesh(40)
    else:
        # twelve triangles with one centroid: a leaf that no split takes apart, cut down by the wide tree's builder
        base = make_soup(2000, seed=9)
        same = np.repeat(base[:1], 12, axis=0).copy()
        same[:, 0:3] += np.linspace(0, 1e-3, 12, dtype=np.float32)[:, None]
        same[:, 3:6] -= np.linspace(0, 1e-3, 12, dtype=np.float32)[:, None] * 0.5
        same[:, 6:9] -= np.linspace(0, 1e-3, 12, dtype=np.float32)[:, None] * 0.5
        tris = np.concatenate([base, same])
    if tris is not None:
        scene = rta.Scene.from_triangles(tris).build_bvh()
    host, dev = both_paths(gpu, scene, monkeypatch)
    for name, a, b in zip(NAMES, host, dev):
        assert a.shape == b.shape, f"{which}: {name} differ in size ({a.shape} / {b.shape})"
        if not np.array_equal(a, b):
            bad = np.nonzero(a != b)[0]
            raise AssertionError(f"{which}: {name} differ in {len(bad)} bytes, first at {bad[0]}")
    assert len(host[4]) > 0, "no wide tree: nothing compared"
    # and the walk over the device-built tree gives the checker's hits
    osc = (orc.Scene.from_triangles(tris) if tris is not None else orc.Scene.load(f"{orc.SCENES}/dragon.txt")).build_bvh()
    nodes = osc.nodes()
    lo, hi = nodes["lo"][0], nodes["hi"][0]
    rr = raygen.random_rays(200_000, lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), seed=21).astype(np.float32)
    got = gpu.trace_closest(rr)
    idx, t, u, v = osc.trace_closest(rr)
    assert np.array_equal(got["prim"], idx)
    assert np.array_equal(got["t"].view(np.uint32), t.view(np.uint32))


def test_what_the_device_path_declines_goes_to_the_host_path(gpu, monkeypatch):
    """cornell_box.txt has a sphere and a leaf of 16 primitives; a leaf of 40 coincident triangles is more
    than one thread cuts down; an invalid description must come back with the host path's wording."""
    monkeypatch.setenv("RTA_UPLOAD_DEVICE", "1")
    scene = rta.Scene.load(f"{orc.SCENES}/cornell_box.txt").build_bvh()
    gpu.upload(scene)
    dev = layouts(gpu)
    monkeypatch.setenv("RTA_UPLOAD_DEVICE", "0")
    gpu.upload(scene)
    host = layouts(gpu)
    for name, a, b in zip(NAMES[:4], host, dev):
        assert np.array_equal(a, b), name
    # a long leaf
    monkeypatch.setenv("RTA_UPLOAD_DEVICE", "1")
    tris = np.concatenate([np.repeat(make_soup(1, seed=2), 40, axis=0), make_soup(500, seed=4)])
    long_leaf = rta.Scene.from_triangles(tris).build_bvh()
    assert long_leaf.arrays()["node_count"].max() >= 40
    gpu.upload(long_leaf)
    rr = raygen.random_rays(50_000, np.full(3, -1.1), np.full(3, 1.1), seed=2).astype(np.float32)
    osc = orc.Scene.from_triangles(tris).build_bvh()
    assert np.array_equal(gpu.trace_closest(rr)["prim"], osc.trace_closest(rr)[0])
    # invalid descriptions: same code and text from either path
    a = scene.arrays()
    for breakage in ("child", "leaf", "twice"):
        b = {k: v.copy() for k, v in a.items()}
        if breakage == "child":
            b["node_index"][0] = len(b["node_index"]) - 1
        elif breakage == "leaf":
            b["node_index"][1] = len(b["prim_type"]) - 1
        else:
            b["node_index"][2] = b["node_index"][1]
        texts = []
        for path in ("0", "1"):
            monkeypatch.setenv("RTA_UPLOAD_DEVICE", path)
            with pytest.raises(rta.RtaError) as e:
                gpu.upload(rta.desc_from_arrays(b))
            texts.append((e.value.code, e.value.text))
        assert texts[0] == texts[1] and texts[0][0] == rta.E_INVALID, (breakage, texts)
    monkeypatch.delenv("RTA_UPLOAD_DEVICE")
    gpu.upload(scene)
