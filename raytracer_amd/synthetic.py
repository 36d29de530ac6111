"""The synthetic intersect-only workload: a seed-fixed random triangle soup and
the two ray batches traced over it (closest hit, rays resident in HBM).

    soup        n triangles, centres uniform in [-1, 1]^3, vertices = centre +
                uniform [-0.01, 0.01]^3 (seed 1)
    coherent    a side x side pinhole grid from (-3, 0, 0) looking at the
                origin, 40 degrees wide, un-jittered
    incoherent  origins uniform in [-1, 1]^3, directions uniform on the sphere
                (seed 2)

Used by bench.py (--workload soup10m), the full-size GPU test and the soup
tools, so that they all trace the same triangles and rays.
"""
import numpy as np


def make_soup(n, seed=1):
    rng = np.random.default_rng(seed)
    out = np.empty((n, 9), np.float32)
    step = 1 << 20
    for a in range(0, n, step):
        m = min(step, n - a)
        c = rng.uniform(-1, 1, (m, 1, 3)).astype(np.float32)
        out[a:a + m] = (c + rng.uniform(-0.01, 0.01, (m, 3, 3)).astype(np.float32)).reshape(m, 9)
    return out


def coherent_rays(side, dev):
    import torch
    half = float(np.tan(np.radians(20.0)))
    t = (torch.arange(side, device=dev, dtype=torch.float32) + 0.5) / side * 2 - 1
    y = (t * half).view(1, side).expand(side, side)
    z = (-t * half).view(side, 1).expand(side, side)
    d = torch.stack([torch.ones_like(y), y, z], dim=-1).reshape(-1, 3)
    d = d / d.norm(dim=1, keepdim=True)
    o = torch.tensor([-3.0, 0.0, 0.0], device=dev).expand_as(d)
    return torch.cat([o, d], dim=1).contiguous()


def incoherent_rays(n, dev, seed=2):
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    o = torch.rand((n, 3), device=dev, generator=g) * 2 - 1
    d = torch.randn((n, 3), device=dev, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    return torch.cat([o, d], dim=1).contiguous()
