"""Camera rays answered from the triangle lists of their cells on the image plane (csrc/cells.hip,
csrc/traverse_cells.hpp) instead of a tree: whole renders against the CPU checker's, bit for bit, from views
that put the lists to work - triangles larger than the frame, seen edge-on, cut by the frame's border, shared
edges aimed at exactly - and the statistics that say the lists were in fact used.  The reference's loop:
src/main.cpp:657-714; its closest hit: src/main.cpp:332-422."""
import re

import numpy as np
import pytest

import orc
import raytracer_amd as rta

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _with_render_block(text, **kw):
    head, sep, tail = text.partition("}")
    for key, val in kw.items():
        pattern = rf"(\b{key}\s+)(\[[^\]]*\]|\S+)"
        if re.search(pattern, head):
            head = re.sub(pattern, lambda m: m.group(1) + val, head, count=1)
        else:
            head += f"\t{key} {val}\n"
    return head + sep + tail


def _vec(v):
    return "[" + " ".join("%r" % float(x) for x in v) + "]"


def _render_both(gpu, text, seed, sps, expect_listed=True):
    osc = orc.Scene.parse(text).build_bvh()
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=seed, samples_per_stream=sps)
    bgr, acc, st = gpu.render_scene(psc, seed=seed, samples_per_stream=sps)
    assert st["rays"] == cn["rays"] and st["samples"] == cn["samples"]
    assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bgr, want_bgr)
    if expect_listed:
        # every camera ray that is traced at all comes from a list (all but rays with a zero component or the like)
        traced_camera = st["samples"] - st["early_rays"]
        assert traced_camera > 0 and 0.99 * traced_camera <= st["listed_rays"] <= traced_camera, st
    else:
        assert st["listed_rays"] == 0, st
    return st


def test_views_around_the_dragon(gpu):
    """dragon.txt from eight places within the wide tree's reach: any distance, any roll, narrow and wide lenses, frames that
    cut the mesh.  Each render equals the checker's."""
    base = open(f"{orc.SCENES}/dragon.txt").read()
    centre, radius = np.array([0.056, 0.0768, 1.172]), 1.644
    rng = np.random.default_rng(11)
    for k in range(8):
        d = rng.standard_normal(3)
        d /= np.linalg.norm(d)
        cam = centre + d * radius * rng.uniform(1.12, 1.2)
        look = centre + rng.uniform(-0.5, 0.5, 3)
        up = rng.standard_normal(3)
        text = _with_render_block(base, camera=_vec(cam), look_at=_vec(look), up=_vec(up), fov="%r" % float(rng.uniform(20, 75)),
                                  width=str(int(rng.integers(40, 130))), height=str(int(rng.integers(30, 90))), samples="6")
        st = _render_both(gpu, text, seed=100 + k, sps=3)
        assert st["walked_rays"] > 0


def test_the_lists_do_not_need_the_wide_tree(gpu):
    """The lists are made from the camera's own distances and each cell's own determinants, not from the wide tree's
    padding: a camera three scene radii away, where the walk does not admit its rays, and cornell_box.txt, whose walls are
    far too large for a wide tree at all (and whose sphere every camera ray tests), are served all the same."""
    base = open(f"{orc.SCENES}/dragon.txt").read()
    text = _with_render_block(base, camera="[-5.5 0.3 1.4]", width="96", height="64", samples="4", fov="25")
    st = _render_both(gpu, text, seed=5, sps=2)
    assert st["walked_rays"] > 0
    text = orc.scene_variant("cornell_box.txt", 96, 72, 6)
    st = _render_both(gpu, text, seed=6, sps=3)
    assert st["walked_rays"] == 0 and st["shadow_rays"] > 0


def _grid(n, z_of, x0=-1.0, x1=1.0):
    tris = []
    h = (x1 - x0) / n
    for i in range(n):
        for j in range(n):
            p = [(x0 + (i + a) * h, x0 + (j + b) * h) for a, b in ((0, 0), (1, 0), (1, 1), (0, 1))]
            q = [(x, y, z_of(x, y)) for x, y in p]
            tris.append(q[0] + q[1] + q[2])
            tris.append(q[0] + q[2] + q[3])
    return np.float32(tris)


def _mesh_text(tris, camera, look_at, up, fov, width, height, samples):
    body = " ".join("%r" % float(x) for x in tris.reshape(-1))
    return (f"render {{\n\tcamera {_vec(camera)}\n\tlook_at {_vec(look_at)}\n\tup {_vec(up)}\n\twidth {width}\n\theight {height}\n"
            f"\tsamples {samples}\n\tfov {fov}\n\tsky_box_color [.5 .6 1]\n}}\n\nmaterial {{\n\tcolor [0.8 0.7 0.6]\n\troughness 1\n}}\n\n"
            f"triangle_mesh {{\n\tvertices [ {body} ]\n}}\n")


def test_triangles_seen_edge_on_and_shared_edges(gpu):
    """A bumpy sheet seen at a grazing angle (its triangles are slivers on the image plane, many of them behind one
    another in a cell's list) and a flat one on a binary grid seen straight on (pixel centres over shared edges: exact
    ties, which the lists hand on to the exact traversal)."""
    bumpy = _grid(24, lambda x, y: 0.11 * np.sin(5 * x) * np.cos(4 * y))
    st = _render_both(gpu, _mesh_text(bumpy, (-1.44, 0.1, 0.16), (0.2, 0, 0), (0, 0, 1), 40, 120, 48, 6), seed=1, sps=3)
    assert st["retraced_rays"] < st["listed_rays"] // 20
    flat = _grid(16, lambda x, y: 0.0)
    st = _render_both(gpu, _mesh_text(flat, (0, 0, 1.45), (0, 0, 0), (0, 1, 0), 64, 64, 64, 8), seed=2, sps=4)
    assert st["retraced_rays"] > 0


def test_triangles_larger_than_the_frame(gpu):
    """A narrow lens on a coarse mesh: most triangles cover many cells, some the whole frame, some are cut by its
    border.  (The cameras of these synthetic scenes stand within 1.04 scene radii of the middle: the reach of a wide tree
    whose boxes are thin.)"""
    coarse = _grid(3, lambda x, y: 0.2 * x * y + 0.1 * x)
    st = _render_both(gpu, _mesh_text(coarse, (0.3, -0.2, 1.4), (0.1, 0.05, 0), (0.2, 1, 0), 9, 80, 60, 6), seed=4, sps=3)
    assert st["early_rays"] < st["samples"]


def _check_lists(gpu, text, w, h, n=2_000_000, seed=7):
    import torch
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    _, _, st = gpu.render_scene(psc, seed=1, samples_per_stream=1)
    assert st["listed_rays"] > 0
    border = 3
    starts = gpu.scene_buffer(7).view(np.uint32)
    prims = gpu.scene_buffer(8).view(np.uint32)
    keys = gpu.scene_buffer(9).view(np.uint64)
    assert starts.size == (w + 2 * border) * (h + 2 * border) + 1 and starts[-1] == prims.size == keys.size > 0
    cam = psc.camera()
    o, ul, du, dv = (np.array(v[:], np.float32) for v in (cam.origin, cam.upper_left, cam.du, cam.dv))
    rng = np.random.default_rng(seed)
    fx = rng.uniform(-2.9, w + 2.9, n).astype(np.float32)
    fy = rng.uniform(-2.9, h + 2.9, n).astype(np.float32)
    target = (ul[None, :] + du[None, :] * fx[:, None]) + dv[None, :] * fy[:, None]
    d = target - o[None, :]
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    rr = np.concatenate([np.broadcast_to(o, (n, 3)), d], axis=1).astype(np.float32)
    dev = torch.device("cuda", 0)
    rt = torch.from_numpy(np.ascontiguousarray(rr)).to(dev)
    ht = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()      # the library works on its own streams
    gpu.trace_variant_device(10, rt.data_ptr(), n, ht.data_ptr())
    got = ht.cpu().numpy()
    hit_prim, hit_t = got[:, 0].view(np.int32), got[:, 1]
    cell = (np.floor(fy).astype(np.int64) + border) * (w + 2 * border) + np.floor(fx).astype(np.int64) + border
    # (spheres are in no list: every camera ray tests them)
    is_triangle = psc.arrays()["prim_type"] == 0
    hits = np.nonzero((hit_prim >= 0) & is_triangle[np.maximum(hit_prim, 0)])[0]
    # every entry as (cell, triangle), sorted: a hit's (cell, triangle) must be one of them
    entry_cell = (keys >> np.uint64(32)).astype(np.uint64)
    assert np.array_equal(entry_cell, np.repeat(np.arange(starts.size - 1, dtype=np.uint64), np.diff(starts.astype(np.int64))))
    entry_bound = (keys & np.uint64(0xffffffff)).astype(np.uint32).view(np.float32)
    combined = (entry_cell << np.uint64(32)) | prims.astype(np.uint64)
    order = np.argsort(combined, kind="stable")
    combined = combined[order]
    want = (cell[hits].astype(np.uint64) << np.uint64(32)) | hit_prim[hits].astype(np.uint64)
    at = np.searchsorted(combined, want)
    found = (at < combined.size) & (combined[np.minimum(at, combined.size - 1)] == want)
    assert found.all(), int((~found).sum())
    assert np.all(entry_bound[order][at] <= hit_t[hits])
    # and the lists are lists, not the whole scene: entries per cell that has any
    lengths = np.diff(starts.astype(np.int64))
    return hits.size, float(lengths[lengths > 0].mean())


def test_every_hit_lies_in_its_cell_s_list(gpu):
    """The lists themselves, read back from the device: two million rays aimed anywhere on the image plane (the frame and
    the filter's reach around it) are traced by the traversal that replays the reference's order; whatever triangle one of
    them hits must be in the list of the cell it was aimed at, with a key (the least distance at which the triangle can be
    hit through the cell) no larger than the distance found.  dragon.txt from two sides, cornell_box.txt (walls seen from
    in front and at a grazing angle: the widening follows each cell's determinant), a bumpy sheet seen edge-on."""
    base = open(f"{orc.SCENES}/dragon.txt").read()
    for cam_at, look, fov in (((-1.9, 0.0, 1.0), (0.0, 0.0, 1.0), 40.0), ((0.9, -1.6, 2.4), (0.0, 0.2, 1.0), 65.0)):
        text = _with_render_block(base, camera=_vec(cam_at), look_at=_vec(look), fov="%r" % fov, width="480", height="270", samples="1")
        n_hits, mean_len = _check_lists(gpu, text, 480, 270)
        assert n_hits > 100_000 and mean_len < 16      # (a cell of this small frame spans several triangles)
    n_hits, mean_len = _check_lists(gpu, orc.scene_variant("cornell_box.txt", 480, 270, 1), 480, 270)
    assert n_hits > 1_000_000 and mean_len < 8
    bumpy = _grid(24, lambda x, y: 0.11 * np.sin(5 * x) * np.cos(4 * y))
    n_hits, mean_len = _check_lists(gpu, _mesh_text(bumpy, (-1.44, 0.1, 0.16), (0.2, 0, 0), (0, 0, 1), 40, 480, 270, 1), 480, 270)
    assert n_hits > 100_000
