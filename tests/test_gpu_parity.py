"""GPU parity: the HIP path against the CPU checker, through the C ABI.

Bars: hit/miss, primitive index, t, u, v and occlusion flags bit-exact;
rendered radiance sums and B,G,R bytes bit-exact against the checker run with
the same per-(pixel, block) streams and the portable elementary functions.
"""
import numpy as np
import pytest

import orc
import rays as raygen
import raytracer_amd as rta

pytestmark = pytest.mark.gpu

SCENES = ("cornell_box.txt", "dragon.txt")


def load_pair(name):
    path = f"{orc.SCENES}/{name}"
    return orc.Scene.load(path).build_bvh(), rta.Scene.load(path).build_bvh()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_device_arithmetic_is_ieee(gpu):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * 10,
                        np.float32([0, -0.0, 1e-39, -1e-39, 1e-45, 3e38, np.inf, -np.inf, np.nan, 1, 3])])
    y = np.concatenate([rng.standard_normal(200000).astype(np.float32),
                        np.float32([0, 0, 3, 1e-3, 2, 1e-3, 2, 0, 1, 3e38, 1e-39])])
    with np.errstate(all="ignore"):
        want = (x / y).astype(np.float32)
        got = gpu.arith("div", x, y)
        ok = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
        assert ok.all(), f"{(~ok).sum()} division mismatches, first {x[~ok][:3]} / {y[~ok][:3]}"
        pos = np.abs(x)
        got = gpu.arith("sqrt", pos)
        want = np.sqrt(pos).astype(np.float32)
        ok = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
        assert ok.all()
        # a*b+a stays unfused, fmaf fuses
        got = gpu.arith("muladd", x, y)
        want = (x * y).astype(np.float32) + x
        ok = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
        assert ok.all(), "contraction is on"
        got = gpu.arith("fma", x, y)
        want = (x.astype(np.float64) * y.astype(np.float64) + x.astype(np.float64)).astype(np.float32)
        ok = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
        assert ok.mean() > 0.9999      # double rounding of the fp64 emulation is the only slack


def test_reciprocal_quotient_is_correctly_rounded(gpu):
    """The slab test's multiply + two fused corrections must equal IEEE division
    wherever the traversal uses it: |d| in [2^-40, 2^40], numerator 0 or of
    magnitude in [2^-78, 2^41]."""
    rng = np.random.default_rng(3)
    n = 4_000_000
    def spread(count, lo_exp, hi_exp):
        mant = rng.integers(0, 1 << 23, count, dtype=np.uint32)
        exp = rng.integers(lo_exp + 127, hi_exp + 127, count, dtype=np.uint32)
        sign = rng.integers(0, 2, count, dtype=np.uint32)
        return ((sign << 31) | (exp << 23) | mant).view(np.float32)
    num = np.concatenate([spread(n, -78, 41), rng.standard_normal(n).astype(np.float32), np.zeros(16, np.float32)])
    den = np.concatenate([spread(n, -40, 40), rng.standard_normal(n).astype(np.float32) + np.float32(1e-6),
                          spread(16, -40, 40)])
    # denominators with an all-ones significand, the classic hard case for reciprocal methods
    ones = (np.uint32(0x7fffff) | (rng.integers(87, 167, 4096, dtype=np.uint32) << 23)).view(np.float32)
    num = np.concatenate([num, spread(4096, -20, 20)])
    den = np.concatenate([den, ones])
    keep = (np.abs(den) >= 2.0 ** -40) & (np.abs(den) <= 2.0 ** 40)
    num, den = num[keep], den[keep]
    want = (num / den).astype(np.float32)
    got = gpu.arith("quot", num, den)
    same = (bits(got) == bits(want)) | ((got == 0) & (want == 0))
    assert same.all(), f"{(~same).sum()} of {same.size} quotients differ, e.g. {num[~same][:3]} / {den[~same][:3]}"


def test_portable_math_matches_cpu(gpu):
    rng = np.random.default_rng(2)
    n = 300000
    ang = (rng.random(n, dtype=np.float32) * np.float32(6.2831855)).astype(np.float32)
    unit = np.concatenate([rng.random(n, dtype=np.float32), np.float32([2 ** -24, 1 - 2 ** -24, 0.5, 1.0])])
    unit = unit[unit > 0]
    base = np.concatenate([rng.random(n, dtype=np.float32), np.float32([0.0031308, 1.0, 1e-6])])
    assert np.array_equal(bits(gpu.arith("sin", ang)), bits(orc.pmath("sin", ang)))
    assert np.array_equal(bits(gpu.arith("cos", ang)), bits(orc.pmath("cos", ang)))
    assert np.array_equal(bits(gpu.arith("log", unit)), bits(orc.pmath("log", unit)))
    for y in (2.0, 5.0, 1 / 2.4):
        yy = np.full_like(base, y)
        assert np.array_equal(bits(gpu.arith("pow", base, yy)), bits(orc.pmath("pow", base, yy))), y
    neg = (base - np.float32(0.5)).astype(np.float32)
    yy = np.full_like(neg, 5.0)
    assert np.array_equal(bits(gpu.arith("pow", neg, yy)), bits(orc.pmath("pow", neg, yy)))


def batches(osc):
    i = osc.info()
    cam = osc.camera()
    nodes = osc.nodes()
    lo, hi = nodes["lo"][0], nodes["hi"][0]
    pad = (hi - lo) * 0.25
    return {
        "camera": raygen.camera_grid(cam, i["width"], i["height"], step=2),
        "incoherent": raygen.random_rays(60000, lo - pad, hi + pad, seed=7),
        "axis": raygen.axis_rays(),
    }


@pytest.mark.parametrize("name", SCENES)
def test_closest_hit_bit_exact(gpu, name):
    osc, psc = load_pair(name)
    gpu.upload(psc)
    for label, rr in batches(osc).items():
        idx, t, u, v = osc.trace_closest(rr)
        hits = gpu.trace_closest(rr)
        assert np.array_equal(hits["prim"], idx), f"{name}/{label}: {(hits['prim'] != idx).sum()} index mismatches"
        hit = idx >= 0
        assert hit.any() or label == "axis"
        assert np.array_equal(bits(hits["t"]), bits(t)), f"{name}/{label}: t differs"
        assert np.array_equal(bits(hits["u"][hit]), bits(u[hit]))
        assert np.array_equal(bits(hits["v"][hit]), bits(v[hit]))
        # the stated tolerance (1e-5 relative) is implied by bit equality; keep it visible
        assert np.allclose(hits["t"][hit], t[hit], rtol=1e-5, atol=0)


@pytest.mark.parametrize("name", SCENES)
def test_occlusion_bit_exact(gpu, name):
    osc, psc = load_pair(name)
    gpu.upload(psc)
    for label, rr in batches(osc).items():
        rng = np.random.default_rng(11)
        dist = (rng.random(rr.shape[0], dtype=np.float32) * np.float32(4.0)).astype(np.float32)
        want = osc.trace_occluded(rr, dist)
        got = gpu.trace_occluded(rr, dist)
        assert np.array_equal(got, want), f"{name}/{label}: {(got != want).sum()} mismatches"
        assert 0 < want.mean() < 1 or label == "axis"


RENDER_CASES = [
    # scene, width, height, spp, samples_per_stream, slots
    ("cornell_box.txt", 64, 36, 8, 4, 0),
    ("cornell_box.txt", 50, 50, 5, 0, 1024),      # one block per pixel, ragged tiles, few slots
    ("cornell_box.txt", 33, 17, 7, 3, 128),       # short last block
    ("dragon.txt", 96, 54, 8, 4, 0),
    ("dragon.txt", 40, 40, 3, 2, 256),
    # many blocks per pixel: a wave's lanes share 32, 16 and 8 pixels (block runs side by side)
    ("cornell_box.txt", 37, 21, 6, 1, 0),         # 6 blocks -> 4 at a time, the second group half empty
    ("dragon.txt", 48, 40, 17, 2, 512),           # 9 blocks -> 8 at a time, ragged last block and group
]


@pytest.mark.parametrize("name,w,h,spp,sps,slots", RENDER_CASES)
def test_render_bit_exact(gpu, name, w, h, spp, sps, slots):
    osc, psc = load_pair(name)
    osc.set_render(w, h, spp)
    psc.set_render(w, h, spp)
    gpu.upload(psc)
    want_bgr, want_acc, want_cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=1234, samples_per_stream=sps)
    bgr, acc, st = gpu.render_scene(psc, seed=1234, samples_per_stream=sps, slots=slots)
    assert st["samples"] == want_cn["samples"] == w * h * spp
    assert st["rays"] == want_cn["rays"]
    assert st["shadow_rays"] == want_cn["shadow_rays"]
    same = bits(acc) == bits(want_acc)
    assert same.all(), f"{(~same).any(axis=2).sum()} of {w * h} pixels differ in radiance bits"
    assert np.array_equal(bgr, want_bgr)


@pytest.mark.parametrize("name,w,h,spp,sps", [("cornell_box.txt", 70, 44, 4, 2), ("dragon.txt", 64, 64, 4, 4)])
def test_shards_assemble_to_the_single_gpu_frame(gpu, name, w, h, spp, sps):
    """Tile sharding is exact: rendering the shards of a 1-, 2-, 3- and 8-way
    split one after the other gives the bits of the unsharded render."""
    osc, psc = load_pair(name)
    psc.set_render(w, h, spp)
    gpu.upload(psc)
    whole_bgr, whole_acc, whole_st = gpu.render_scene(psc, seed=5, samples_per_stream=sps)
    for n in (2, 3, 8):
        acc = np.zeros_like(whole_acc)
        bgr = np.zeros_like(whole_bgr)
        rays = 0
        for r in range(n):
            b, a, st = gpu.render_scene(psc, seed=5, samples_per_stream=sps, shard_index=r, shard_count=n)
            px = rta.shard_pixels(w, h, r, n)
            untouched = np.ones(w * h, bool)
            untouched[px] = False
            assert not a.reshape(-1, 3)[untouched].any(), "a shard wrote pixels it does not own"
            acc += a
            bgr += b
            rays += st["rays"]
        assert np.array_equal(bits(acc), bits(whole_acc)), n
        assert np.array_equal(bgr, whole_bgr), n
        assert rays == whole_st["rays"]
    # the resolve entry point reproduces the bytes from the gathered sums
    assert np.array_equal(gpu.resolve(whole_acc, spp, psc.settings.exposure), whole_bgr)


@pytest.mark.parametrize("name,w,h,spp,sps", [("cornell_box.txt", 61, 35, 24, 2), ("dragon.txt", 64, 48, 20, 3)])
def test_sample_ranges_sum_to_the_whole_render(gpu, name, w, h, spp, sps):
    """Sample-range sharding: every rank renders all pixels for a range of the
    sample blocks and the partial sums are added (a sum-reduce).  The ranges
    together draw exactly the samples of the unsplit render, so ray counts add
    up exactly and the summed frame equals the unsplit one up to the order of
    the last fp32 additions: stated tolerance 4 ulp of the pixel sum per
    channel, and the resolved bytes may differ by one step in a few pixels."""
    osc, psc = load_pair(name)
    psc.set_render(w, h, spp)
    gpu.upload(psc)
    s = psc.settings
    whole_bgr, whole, whole_st = gpu.render_scene(psc, seed=31, samples_per_stream=sps)
    n_blocks = -(-spp // sps)
    for cuts in ([0, n_blocks // 2, n_blocks], [0, 1, 2, n_blocks - 1, n_blocks], [0, n_blocks]):
        total = np.zeros_like(whole, dtype=np.float32)
        rays = samples = 0
        for b0, b1 in zip(cuts[:-1], cuts[1:]):
            _, part, st = gpu.render(psc.camera(), w, h, spp, exposure=s.exposure, sky=tuple(s.sky), seed=31,
                                     samples_per_stream=sps, block_begin=b0, block_end=b1, want_bgr=False)
            total = (total + part).astype(np.float32)
            rays += st["rays"] + st["shadow_rays"]
            samples += st["samples"]
        assert samples == w * h * spp and rays == whole_st["rays"] + whole_st["shadow_rays"], cuts
        if len(cuts) == 2:
            assert np.array_equal(bits(total), bits(whole))            # one range holding every block is the whole render
        tol = 4 * np.spacing(np.maximum(np.abs(whole), np.float32(1e-6)).astype(np.float32))
        assert (np.abs(total - whole) <= tol).all(), cuts
        got = gpu.resolve(total, spp, s.exposure)
        assert (np.abs(got.astype(np.int32) - whole_bgr.astype(np.int32)) <= 1).all()
        assert (got != whole_bgr).mean() < 0.01
    with pytest.raises(rta.RtaError):
        gpu.render(psc.camera(), w, h, spp, samples_per_stream=sps, block_begin=0, block_end=n_blocks + 1, want_bgr=False)
    with pytest.raises(rta.RtaError):       # a partial sum cannot be tone-mapped
        gpu.render(psc.camera(), w, h, spp, samples_per_stream=sps, block_begin=0, block_end=1)


@pytest.mark.parametrize("sps,world", [(0, 2), (4, 5), (3, 8)])
def test_more_ranks_than_sample_blocks(gpu, sps, world):
    """Sample-range sharding with ranks that get no block (one block per pixel
    split two ways; two blocks over five ranks; three over eight): such a rank
    must contribute zeros, not the whole frame a second time (block range 0, 0
    means "all blocks" to rta_render).  The shares of raytracer_amd.dist summed as
    its all-reduce would sum them give the unsplit frame."""
    import torch
    from raytracer_amd.dist import block_ranges, render_share
    w, h, spp = 40, 24, 8
    osc, psc = load_pair("cornell_box.txt")
    psc.set_render(w, h, spp)
    gpu.upload(psc)
    s = psc.settings
    _, whole, whole_st = gpu.render_scene(psc, seed=5, samples_per_stream=sps)
    shares = block_ranges(spp, sps, world)
    assert sum(r is None for r in shares) == world - (1 if sps <= 0 else -(-spp // sps))
    dev = torch.device("cuda", 0)
    total = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    samples = 0
    for share in shares:
        frame = torch.full((h, w, 3), 7.0, dtype=torch.float32, device=dev)      # (stale contents must not survive)
        torch.cuda.synchronize()
        p = rta.RenderParams(width=w, height=h, samples=spp, exposure=s.exposure, seed=5, samples_per_stream=sps,
                             shard_index=0, shard_count=1, out_mem=rta.MEM_DEVICE)
        p.sky[:] = list(s.sky)
        st = render_share(gpu, psc.camera(), p, share, frame)
        torch.cuda.synchronize()
        total += frame
        samples += st["samples"]
    assert samples == whole_st["samples"] == w * h * spp
    got = total.cpu().numpy()
    tol = 4 * np.spacing(np.maximum(np.abs(whole), np.float32(1e-6)).astype(np.float32))
    assert (np.abs(got - whole) <= tol).all()


def test_scheduling_knobs_do_not_change_the_image(gpu):
    osc, psc = load_pair("cornell_box.txt")
    psc.set_render(40, 24, 6)
    gpu.upload(psc)
    base = gpu.render_scene(psc, seed=9, samples_per_stream=2)[1]
    for kw in (dict(slots=64), dict(slots=100000, streams=1), dict(streams=8, run_blocks=3), dict(run_blocks=1, streams=2),
               dict(flags=rta.RENDER_SMALL_PASSES), dict(flags=rta.RENDER_SMALL_PASSES, run_blocks=2, slots=256)):
        other = gpu.render_scene(psc, seed=9, samples_per_stream=2, **kw)[1]
        assert np.array_equal(bits(other), bits(base)), kw


def test_edge_inputs(gpu):
    osc, psc = load_pair("cornell_box.txt")
    gpu.upload(psc)
    assert len(gpu.trace_closest(np.zeros((0, 6), np.float32))) == 0
    # NaN and zero directions must come back as misses, not hang or fault
    weird = np.float32([[0, 0, 1, 0, 0, 0], [0, 0, 1, np.nan, 0, 1], [np.inf, 0, 1, 1, 0, 0], [0, 0, 1, 1e-30, 1, 0]])
    want = osc.trace_closest(weird)[0]
    assert np.array_equal(gpu.trace_closest(weird)["prim"], want)
    with pytest.raises(rta.RtaError):
        gpu.render(psc.camera(), 0, 10, 1)
    with pytest.raises(rta.RtaError):
        gpu.render(psc.camera(), 8, 8, 1, shard_index=2, shard_count=2)
    fresh = rta.Context(0)
    with pytest.raises(rta.RtaError) as e:
        fresh.trace_closest(weird)
    assert e.value.code == rta.E_NO_SCENE
    fresh.close()


def test_single_pixel_and_single_sample(gpu):
    osc, psc = load_pair("dragon.txt")
    for w, h, spp in ((1, 1, 1), (1, 9, 2), (9, 1, 1)):
        osc.set_render(w, h, spp)
        psc.set_render(w, h, spp)
        gpu.upload(psc)
        want = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=2, samples_per_stream=0)[1]
        got = gpu.render_scene(psc, seed=2)[1]
        assert np.array_equal(bits(got), bits(want)), (w, h, spp)


def test_features_unused_by_the_shipped_scenes(gpu):
    """Defocus lens sampling, exposure, a rough metal and a second sphere."""
    text = orc.scene_variant("cornell_box.txt", 56, 40, 5).replace(
        "look_at [0 0 1]", "look_at [0 0 1]\n\tdefocus_angle 2.5\n\texposure 1.7\n\tfov 70")
    text += "\nmaterial { color [0.9 0.8 0.3] roughness 0.35 metallic 1 ior 1.8 }\nsphere { center [0.3 -0.4 0.25] radius 0.25 }\n"
    osc = orc.Scene.parse(text).build_bvh()
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    want_bgr, want_acc, _ = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=21, samples_per_stream=2)
    bgr, acc, _ = gpu.render_scene(psc, seed=21, samples_per_stream=2)
    assert np.array_equal(bits(acc), bits(want_acc))
    assert np.array_equal(bgr, want_bgr)


@pytest.mark.parametrize("angle,look_at", [(2.0, "[0 0 1]"), (0.7, "[2 0.2 1.1]")])
def test_lens_camera_on_the_dragon_bit_exact(gpu, angle, look_at):
    """dragon.txt through a lens (focus on the mesh, and far behind it): most
    samples around the blurred silhouette are decided on the image plane, where
    a lens point sees a box shifted by (1 - h / depth) times its offset; all
    480 x 270 x 64 samples must agree with the checker's full computation."""
    text = orc.scene_variant("dragon.txt", 480, 270, 64).replace("look_at [0 0 1]", f"look_at {look_at}\n\tdefocus_angle {angle}")
    assert "defocus_angle" in text
    osc = orc.Scene.parse(text).build_bvh()
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    want_bgr, want_acc, cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=5, samples_per_stream=8)
    bgr, acc, st = gpu.render_scene(psc, seed=5, samples_per_stream=8)
    assert st["rays"] == cn["rays"]
    assert st["early_rays"] > 0.5 * st["rays"]
    # the rest - rays from the lens as well as bounce rays - take the walk over the wide tree
    assert st["walked_rays"] > 0.99 * (st["rays"] - st["early_rays"]), st
    same = bits(acc) == bits(want_acc)
    assert same.all(), f"{(~same).any(axis=2).sum()} pixels differ in radiance bits"
    assert np.array_equal(bgr, want_bgr)


def _box_mesh(lo, hi):
    """12 triangles of the axis-parallel box lo..hi as a triangle_mesh block."""
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    c = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
    quads = [(0, 1, 2, 3), (4, 7, 6, 5), (0, 4, 5, 1), (1, 5, 6, 2), (2, 6, 7, 3), (3, 7, 4, 0)]
    rows = []
    for a, b, cc, d in quads:
        for tri in ((a, b, cc), (a, cc, d)):
            rows.append(" ".join(f"{v:g}" for i in tri for v in c[i]))
    return "triangle_mesh {\n\tvertices [\n\t\t" + "\n\t\t".join(rows) + "\n\t]\n}\n"


_SPHERES = ("sphere {\n\tcenter [0 -0.3 0.1]\n\tradius 0.45\n}\n\nsphere {\n\tcenter [0.2 0.5 -0.2]\n\tradius 0.3\n}\n\n"
            "sphere {\n\tcenter [-0.4 0.1 0.6]\n\tradius 0.2\n}\n")


@pytest.mark.parametrize("camera,look_at,geometry,lens", [
    ("[-3 0 0.1]", "[0 0 0.1]", "box", ""),          # seen along an axis: the outline is the projected box on all four sides
    ("[-2.5 1.3 0.9]", "[0 0.1 0]", "box", ""),
    ("[-2.2 0.4 0.3]", "[0 0 0.1]", "spheres", ""),
    ("[-2.5 1.3 0.9]", "[0 1.1 0.9]", "box", ""),    # the mesh is cut by the edge of the frame
    # lens cameras: the outline is blurred by the lens; focus in front of, inside and behind the mesh
    ("[-3 0 0.1]", "[0 0 0.1]", "box", "defocus_angle 3"),
    ("[-2.5 1.3 0.9]", "[-1.5 0.8 0.55]", "box", "defocus_angle 1.5"),
    ("[-2.2 0.4 0.3]", "[3 -0.5 0]", "spheres", "defocus_angle 0.8"),
])
def test_samples_answered_on_the_image_plane(gpu, camera, look_at, geometry, lens):
    """Camera samples whose pixel-filter radius cannot reach the projection of
    any leaf's box are given the sky without a ray.  A box-shaped mesh fills
    its own bounding box and a sphere touches all six faces of its own, so the
    pixels around their outlines mix samples decided either way, and all must
    agree with the checker's full computation."""
    text = ("render {\n\tcamera %s\n\tlook_at %s\n\t%s\n\tsamples 96\n\twidth 44\n\theight 40\n"
            "\tsky_box_color [.5 .6 1]\n}\n\nmaterial {\n\tcolor [0.8 0.3 0.2]\n\tior 1.5\n\troughness 1\n\tmetallic 0\n}\n\n"
            % (camera, look_at, lens)) + (_box_mesh((-0.5, -0.62, -0.41), (0.5, 0.57, 0.63)) if geometry == "box" else _SPHERES)
    osc = orc.Scene.parse(text).build_bvh()
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    want_bgr, want_acc, want_cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=77, samples_per_stream=8)
    bgr, acc, st = gpu.render_scene(psc, seed=77, samples_per_stream=8)
    assert st["rays"] == want_cn["rays"]
    assert 0 < st["early_rays"] < st["rays"]          # some samples took the short way, not all
    same = bits(acc) == bits(want_acc)
    assert same.all(), f"{(~same).any(axis=2).sum()} pixels differ in radiance bits"
    assert np.array_equal(bgr, want_bgr)


def test_triangle_soup_indices_bit_exact(gpu):
    """The intersect-only workload at test size: 200 k random triangles, a
    coherent pinhole grid and incoherent rays; every index and t must match."""
    rng = np.random.default_rng(1)
    n = 200_000
    c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
    tri = (c + rng.uniform(-0.01, 0.01, (n, 3, 3)).astype(np.float32)).reshape(n, 9)
    osc = orc.Scene.from_triangles(tri).build_bvh()
    psc = rta.Scene.from_triangles(tri).build_bvh()
    gpu.upload(psc)
    osc.set_camera((-3, 0, 0), (0, 0, 0), fov=40.0).set_render(512, 512, 1)
    coherent = raygen.camera_grid(osc.camera(), 512, 512)
    incoherent = raygen.random_rays(300_000, (-1, -1, -1), (1, 1, 1), seed=3)
    for label, rr in (("coherent", coherent), ("incoherent", incoherent)):
        idx, t, u, v = osc.trace_closest(rr)
        hits = gpu.trace_closest(rr)
        assert np.array_equal(hits["prim"], idx), label
        assert np.array_equal(bits(hits["t"]), bits(t)), label
        assert np.array_equal(bits(hits["u"]), bits(u)) and np.array_equal(bits(hits["v"]), bits(v)), label
        assert 0.05 < (idx >= 0).mean() < 0.999, label
        # occlusion queries over the same rays, cut off at and shortly before the hits found above (the
        # incoherent batch is big enough to be traced in the order of its origins, the grid comes in order)
        tmax = np.where(idx >= 0, t * np.float32(1.0 if label == "coherent" else 0.999), np.float32(2.5)).astype(np.float32)
        assert np.array_equal(gpu.trace_occluded(rr, tmax), osc.trace_occluded(rr, tmax)), label


@pytest.mark.parametrize("name,spp", [("dragon.txt", 8), ("cornell_box.txt", 2)])
def test_full_resolution_frames_bit_exact(gpu, name, spp):
    """1920x1080 (the benchmark's frame size) at a sample count the CPU checker
    finishes in seconds: every one of the 2 M pixels must carry the checker's
    radiance bits and bytes."""
    osc, psc = load_pair(name)
    osc.set_render(1920, 1080, spp)
    psc.set_render(1920, 1080, spp)
    gpu.upload(psc)
    want_bgr, want_acc, want_cn = osc.render(orc.RNG_STREAM, orc.MATH_PM, seed=69420, samples_per_stream=4)
    bgr, acc, st = gpu.render_scene(psc, seed=69420, samples_per_stream=4)
    assert st["rays"] == want_cn["rays"] and st["shadow_rays"] == want_cn["shadow_rays"]
    assert np.array_equal(bits(acc), bits(want_acc))
    assert np.array_equal(bgr, want_bgr)


def test_benchmark_size_properties(gpu):
    """dragon 1080p at 256 spp (a quarter of the benchmark's samples, same
    kernels and queues at full occupancy): sky pixels resolve to the exact sky
    bytes, the frame agrees with a low-sample frame of different blocks within
    Monte-Carlo noise, and splitting the samples into blocks differently moves
    only the noise."""
    osc, psc = load_pair("dragon.txt")
    psc.set_render(1920, 1080, 256)
    gpu.upload(psc)
    bgr, acc, st = gpu.render_scene(psc, seed=69420, samples_per_stream=32)
    assert st["samples"] == 1920 * 1080 * 256
    assert 1.15 < st["rays"] / st["samples"] < 1.26            # 1.206 in the reference's own traversal
    # a pixel all of whose 256 samples left the scene sums to exactly 256 * sky (8 block sums of 32 samples)
    pure_sky = (acc == np.float32([128, 128, 256])).all(axis=2)
    assert 0.87 < pure_sky.mean() < 0.895                      # 89.0 % of the frame is sky at 1080p
    assert (bgr[pure_sky] == np.uint8([187, 156, 156])).all()  # B,G,R bytes of the sky colour (.5,.5,1)
    psc.set_render(1920, 1080, 16)
    low = gpu.render_scene(psc, seed=7, samples_per_stream=16)[0].astype(np.float64)
    rmse = float(np.sqrt(np.mean((low - bgr.astype(np.float64)) ** 2)))
    assert rmse < 2.5, rmse                                    # noise of the 16-spp frame: ~1.5 expected


def test_design_variants_return_the_same_bits(gpu):
    """The measurement-only traversal variants (private-memory stack, division
    box test, reference node layout) must agree with the production kernel."""
    import torch
    osc, psc = load_pair("dragon.txt")
    gpu.upload(psc)
    rr = np.concatenate([raygen.camera_grid(osc.camera(), 400, 400, step=2),
                         raygen.random_rays(50000, (-0.6, -1.3, 0.2), (0.7, 1.4, 2.1), seed=9), raygen.axis_rays()])
    want = gpu.trace_closest(rr)
    dev = torch.device("cuda", 0)
    rt = torch.from_numpy(np.ascontiguousarray(rr)).to(dev)
    for variant in (0, 1, 2, 3, 4, 5, 101, 124):
        ht = torch.zeros((rr.shape[0], 4), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        gpu.trace_variant_device(variant, rt.data_ptr(), rr.shape[0], ht.data_ptr())
        got = ht.cpu().numpy()
        assert np.array_equal(got[:, 0].view(np.int32), want["prim"]), variant
        assert np.array_equal(got[:, 1].view(np.uint32), want["t"].view(np.uint32)), variant


def _same_tree(a, b):
    """Every array of the built scene, bit for bit (boxes compared as bits: signed zeros count)."""
    aa, bb = a.arrays(), b.arrays()
    for key in aa:
        x, y = aa[key], bb[key]
        if x.dtype == np.float32:
            x, y = x.view(np.uint32), y.view(np.uint32)
        assert np.array_equal(x, y), key
    assert np.array_equal(a.permutation(), b.permutation())


def _same_as_checker(scene, checker):
    """The built scene against the CPU checker's own build of the same input:
    node records, primitive order and the light links, bit for bit."""
    a = scene.arrays()
    nodes, objs = checker.nodes(), checker.objects()
    assert len(nodes) == len(a["node_index"]) and len(objs) == len(a["prim_type"])
    assert np.array_equal(a["node_index"], nodes["index"]) and np.array_equal(a["node_count"], nodes["count"])
    assert np.array_equal(bits(a["node_bounds"][:, 0:3]), bits(nodes["lo"]))
    assert np.array_equal(bits(a["node_bounds"][:, 3:6]), bits(nodes["hi"]))
    assert np.array_equal(a["prim_type"], objs["type"])
    assert np.array_equal(a["prim_material"], objs["material"]) and np.array_equal(a["prim_light"], objs["light"])
    tri = a["prim_type"] == rta.PRIM_TRIANGLE
    assert np.array_equal(bits(a["prim_geom"][tri]), bits(objs["geom"][tri, 0:9]))
    assert np.array_equal(bits(a["prim_normals"][tri]), bits(objs["geom"][tri, 9:18]))
    assert np.array_equal(bits(a["prim_geom"][~tri, 0:4]), bits(objs["geom"][~tri, 0:4]))
    assert np.array_equal(a["light_prim"], checker.lights()["object"])


def test_device_bvh_build_is_the_host_tree(gpu):
    """The GPU builder reproduces the reference builder's tree exactly: node
    numbering, boxes, primitive order, light links - on the shipped scenes, on
    spheres, on inputs whose splits fail, on signed zeros and on a 300 k soup.
    Every device-built tree is compared with the CPU checker's build of the same
    input and with the library's host builder."""
    for name in ("cornell_box.txt", "dragon.txt"):
        path = f"{orc.SCENES}/{name}"
        host = rta.Scene.load(path).build_bvh()
        dev = rta.Scene.load(path)
        dev.build_bvh_on(gpu)
        _same_tree(host, dev)
        _same_as_checker(dev, orc.Scene.load(path).build_bvh())
    rng = np.random.default_rng(3)
    cases = {
        "one": rng.uniform(-1, 1, (1, 9)),
        "two": rng.uniform(-1, 1, (2, 9)),
        "three": rng.uniform(-1, 1, (3, 9)),
        # identical centroids: every split fails after permuting the range
        "coincident": np.tile(rng.uniform(-1, 1, (1, 9)), (37, 1)),
        # half of the centroids coincide: large leaves next to ordinary subtrees
        "half coincident": np.concatenate([np.tile(rng.uniform(-1, 1, (1, 9)), (500, 1)), rng.uniform(-1, 1, (700, 9))]),
        # coordinates that are +0 and -0: the boxes must carry the same sign bits
        "signed zeros": np.where(rng.random((4000, 9)) < 0.3, np.where(rng.random((4000, 9)) < 0.5, 0.0, -0.0),
                                 rng.uniform(-1, 1, (4000, 9))),
        # a grid: many equal coordinates and centroids exactly on split planes
        "grid": (rng.integers(0, 8, (5000, 1, 3)) + rng.integers(0, 2, (5000, 3, 3))).reshape(5000, 9) / 4.0,
        "soup": (rng.uniform(-1, 1, (300_000, 1, 3)) + rng.uniform(-0.01, 0.01, (300_000, 3, 3))).reshape(-1, 9),
    }
    for label, tri in cases.items():
        tri = np.ascontiguousarray(tri, np.float32)
        host = rta.Scene.from_triangles(tri).build_bvh()
        dev = rta.Scene.from_triangles(tri)
        ms = dev.build_bvh_on(gpu)
        assert ms > 0, label
        try:
            _same_tree(host, dev)
            _same_as_checker(dev, orc.Scene.from_triangles(tri).build_bvh())
        except AssertionError as e:
            raise AssertionError(f"{label}: {e}")
    # spheres and triangles mixed, through the scene grammar
    text = orc.scene_variant("cornell_box.txt", 32, 32, 1)
    for k in range(40):
        c = rng.uniform(-1, 1, 3)
        text += f"\nsphere {{ center [{c[0]:.5f} {c[1]:.5f} {c[2]:.5f}] radius {rng.uniform(0.01, 0.2):.5f} }}\n"
    host = rta.Scene.parse(text).build_bvh()
    dev = rta.Scene.parse(text)
    dev.build_bvh_on(gpu)
    _same_tree(host, dev)
    _same_as_checker(dev, orc.Scene.parse(text).build_bvh())
    # a tree built on the device renders like the host's
    gpu.upload(dev)
    a = gpu.render_scene(dev, seed=5)[1]
    gpu.upload(host)
    b = gpu.render_scene(host, seed=5)[1]
    assert np.array_equal(bits(a), bits(b))
    # non-finite coordinates go to the host builder
    bad = np.ascontiguousarray(rng.uniform(-1, 1, (10, 9)), np.float32)
    bad[3, 4] = np.nan
    host = rta.Scene.from_triangles(bad).build_bvh()
    dev = rta.Scene.from_triangles(bad)
    assert dev.build_bvh_on(gpu) == 0
    _same_tree(host, dev)


def test_cli_writes_the_same_image(gpu, tmp_path):
    """`raytracer FILE` -> ./image.bmp: the file the library path would write
    (the CLI builds its tree on the GPU, this test on the host)."""
    import subprocess
    text = orc.scene_variant("cornell_box.txt", 52, 30, 6)
    (tmp_path / "scene.txt").write_text(text)
    p = subprocess.run([rta.CLI_PATH, "scene.txt", "--seed", "77", "--samples-per-stream", "4"], cwd=tmp_path,
                       stderr=subprocess.PIPE, text=True, timeout=100)
    assert p.returncode == 0, p.stderr
    psc = rta.Scene.parse(text).build_bvh()
    gpu.upload(psc)
    bgr = gpu.render_scene(psc, seed=77, samples_per_stream=4)[0]
    rta.write_bmp(str(tmp_path / "want.bmp"), bgr)
    assert (tmp_path / "image.bmp").read_bytes() == (tmp_path / "want.bmp").read_bytes()
    # (how far the image may be from the reference binary's own render is stated and checked in
    # test_gpu_full_size.py::test_image_within_tolerance_of_the_reference_render)
