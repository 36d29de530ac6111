"""rta_trace_closest_static, the entry point that allocates nothing: its hits
against rta_trace_closest's on the same device-resident rays, bit for bit, on
two calls in a row (the second with a smaller batch, so that the shared static
workspace still holds the first call's keys and order), with the GPU's free
memory unchanged across the calls, and its size limit."""
import numpy as np
import pytest
import torch

import rays as raygen
import raytracer_amd as rta

pytestmark = pytest.mark.gpu


def test_static_workspace_entry_point_matches_and_allocates_nothing(gpu):
    rng = np.random.default_rng(5)
    n_tri = 200_000                       # a tree of 17 MB: rta_trace_closest sorts such batches too
    c = rng.uniform(-1, 1, (n_tri, 1, 3)).astype(np.float32)
    tri = (c + rng.uniform(-0.01, 0.01, (n_tri, 3, 3)).astype(np.float32)).reshape(n_tri, 9)
    scene = rta.Scene.from_triangles(tri).build_bvh()
    gpu.upload(scene)
    dev = torch.device("cuda", 0)
    batches = [raygen.random_rays(700_000, (-1, -1, -1), (1, 1, 1), seed=31),
               raygen.random_rays(90_001, (-1.2, -1.2, -1.2), (1.2, 1.2, 1.2), seed=32)]
    rays = [torch.from_numpy(b).to(dev) for b in batches]
    want = [torch.zeros((len(b), 4), dtype=torch.float32, device=dev) for b in batches]
    got = [torch.full((len(b), 4), -7.0, dtype=torch.float32, device=dev) for b in batches]
    torch.cuda.synchronize()
    for r, h in zip(rays, want):
        gpu.trace_closest_device(r.data_ptr(), r.shape[0], h.data_ptr())
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for r, h in zip(rays, got):           # two calls in a row, the second smaller
        gpu.trace_closest_static_device(r.data_ptr(), r.shape[0], h.data_ptr())
    assert torch.cuda.mem_get_info()[0] == free_before, "device memory was allocated or freed"
    for k, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"call {k}: hits differ from rta_trace_closest"
        hit = (b[:, 0].view(torch.int32) >= 0).float().mean().item()
        assert 0.1 < hit < 1.0
    # the first batch again after the smaller one
    again = torch.zeros_like(got[0])
    gpu.trace_closest_static_device(rays[0].data_ptr(), rays[0].shape[0], again.data_ptr())
    assert torch.equal(again.view(torch.int32), want[0].view(torch.int32))
    # empty and oversized batches
    gpu.trace_closest_static_device(rays[0].data_ptr(), 0, again.data_ptr())
    with pytest.raises(rta.RtaError) as e:
        gpu.trace_closest_static_device(rays[0].data_ptr(), rta.STATIC_MAX_RAYS + 1, again.data_ptr())
    assert e.value.code == rta.E_LIMIT and "at most" in e.value.text
    # a second context on the same GPU finds the same workspace
    other = rta.Context(0)
    other.upload(scene)
    other.trace_closest_static_device(rays[1].data_ptr(), rays[1].shape[0], again.data_ptr())
    assert torch.equal(again[: rays[1].shape[0]].view(torch.int32), want[1].view(torch.int32))
    other.close()
