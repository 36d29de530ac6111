"""Deterministic ray batches shared by the CPU and GPU parity tests."""
import numpy as np


def camera_grid(cam19, width, height, step=1):
    """Un-jittered pixel-centre rays of a pinhole camera given as the 19 floats
    {origin, upper_left, du, dv, lens_u, lens_v, defocus}."""
    cam19 = np.asarray(cam19, np.float32)
    o, ul, du, dv = cam19[0:3], cam19[3:6], cam19[6:9], cam19[9:12]
    us = np.arange(0, width, step, dtype=np.float32) + np.float32(0.5)
    vs = np.arange(0, height, step, dtype=np.float32) + np.float32(0.5)
    uu, vv = np.meshgrid(us, vs)
    target = ul[None, None, :] + du[None, None, :] * uu[..., None] + dv[None, None, :] * vv[..., None]
    d = (target - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    rays = np.concatenate([np.broadcast_to(o, d.shape), d], axis=-1).reshape(-1, 6)
    return np.ascontiguousarray(rays, np.float32)


def random_rays(n, lo, hi, seed):
    """Origins uniform in the box [lo, hi], directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    lo = np.asarray(lo, np.float32)
    hi = np.asarray(hi, np.float32)
    o = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([o, d], axis=1), np.float32)


def axis_rays():
    """Rays with zero direction components (0/0 and x/0 in the slab test)."""
    out = []
    for o in ((-1.9, 0, 1), (0, 0, 1), (0.3, 0.2, 0.5), (-0.99, -1, 0), (1.04, 0.99, 0), (0, 0, 3)):
        for d in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (0.70710677, 0.70710677, 0),
                  (0, -0.70710677, 0.70710677), (-0.0, 1, -0.0)):
            out.append(list(o) + list(d))
    return np.asarray(out, np.float32)
