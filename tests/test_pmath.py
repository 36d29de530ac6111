"""The portable elementary functions (include/rt_pmath.h) against binary64
libm: within 1 ulp of the correctly rounded binary32 value (in practice equal
except in a handful of cases per million)."""
import numpy as np

import orc


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)


def test_sin_cos():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 400000), rng.uniform(-50, 50, 100000),
                        [0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi, 1e-8, 6.2831855]]).astype(np.float32)
    for name, fn in (("sin", np.sin), ("cos", np.cos)):
        got = orc.pmath(name, x)
        want = fn(x.astype(np.float64)).astype(np.float32)
        d = ulp_diff(got, want)
        big = np.abs(want) > 1e-6          # near a zero of the function one ulp of the result is tiny
        assert d[big].max() <= 1, name
        assert np.abs(got - want)[~big].max() < 1e-12
        assert (d[big] == 0).mean() > 0.9999


def test_sincos_pair_has_the_bits_of_the_single_functions():
    """pm_sincosf is what the kernels call where the reference calls cosf and
    sinf of one angle; the checker calls pm_sinf and pm_cosf."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 500000), rng.uniform(-1000, 1000, 200000),
                        [0, -0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi, 1e-8, 6.2831855, 1e6, np.inf, np.nan]]
                       ).astype(np.float32)
    assert np.array_equal(orc.pmath("sincos_s", x).view(np.uint32), orc.pmath("sin", x).view(np.uint32))
    assert np.array_equal(orc.pmath("sincos_c", x).view(np.uint32), orc.pmath("cos", x).view(np.uint32))


def test_log():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.random(400000), np.arange(1, 4097) / 2 ** 24, rng.uniform(1, 1e6, 50000),
                        [1.0, 0.5, 2.0, 1e-38, 1e-44, 3e38]]).astype(np.float32)
    got = orc.pmath("log", x)
    want = np.log(x.astype(np.float64)).astype(np.float32)
    d = ulp_diff(got, want)
    assert d.max() <= 1 and (d == 0).mean() > 0.9999
    with np.errstate(all="ignore"):
        assert orc.pmath("log", np.float32([0]))[0] == -np.inf
        assert np.isnan(orc.pmath("log", np.float32([-1]))[0])


def test_pow():
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.random(300000), rng.uniform(0, 30, 50000), [0.0031308, 1.0, 0.5]]).astype(np.float32)
    for y in (1 / 2.4, 0.3, 1.7, 3.0):
        yy = np.full_like(x, np.float32(y))
        got = orc.pmath("pow", x, yy)
        want = np.power(x.astype(np.float64), np.float64(np.float32(y))).astype(np.float32)
        d = ulp_diff(got, want)
        assert d.max() <= 1 and (d == 0).mean() > 0.999, y
    two = orc.pmath("pow", x, np.full_like(x, 2))
    assert np.array_equal(two, x * x)                      # y == 2 is the correctly rounded square
    xs = (x - np.float32(0.5)).astype(np.float32)
    five = orc.pmath("pow", xs, np.full_like(xs, 5))
    want = (xs.astype(np.float64) ** 5).astype(np.float32)
    assert ulp_diff(five, want).max() <= 1 and (five < 0).any()
    assert orc.pmath("pow", np.float32([0, 7]), np.float32([0.4, 0]))[0] == 0
    assert orc.pmath("pow", np.float32([0, 7]), np.float32([0.4, 0]))[1] == 1
