"""The N > 1 path on CPU: world_size 2 and 3 over gloo.  Each rank owns the
tiles rta_shard_pixels gives it, fills them from the CPU checker's render (the
GPU is not available here), and the shared gather code must reproduce the
single-process frame bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orc
import raytracer_amd as rta
from raytracer_amd.dist import ShardGather, block_ranges, reduce_partial_sums

W, H, SPP = 52, 37, 3          # ragged: neither dimension is a multiple of the tile


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = orc.Scene.load(os.path.join(orc.SCENES, "cornell_box.txt")).build_bvh().set_render(W, H, SPP)
        _, full, _ = sc.render(orc.RNG_STREAM, orc.MATH_PM, seed=11, samples_per_stream=2, threads=2)
        mine = rta.shard_pixels(W, H, rank, world)
        frame = torch.zeros((H, W, 3), dtype=torch.float32)
        frame.view(-1, 3)[torch.from_numpy(mine.astype(np.int64))] = torch.from_numpy(full.reshape(-1, 3)[mine])
        g = ShardGather(W, H, rank, world, torch.device("cpu"))
        g.gather(frame, dist)
        assert np.array_equal(frame.numpy().view(np.uint32), full.view(np.uint32)), f"rank {rank}: frame differs"
        np.save(os.path.join(out_dir, f"frame{rank}.npy"), frame.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_process(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    frames = [np.load(tmp_path / f"frame{r}.npy") for r in range(world)]
    for f in frames[1:]:
        assert np.array_equal(f.view(np.uint32), frames[0].view(np.uint32))


def test_gather_is_identity_for_one_rank():
    frame = torch.arange(H * W * 3, dtype=torch.float32).reshape(H, W, 3)
    g = ShardGather(W, H, 0, 1, torch.device("cpu"))
    assert g.gather(frame.clone(), None).equal(frame)
    assert g.counts == [W * H]


def _reduce_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank contributes a frame that depends on its block range; the sum must be the same everywhere
        b0, b1 = block_ranges(37, 4, world)[rank]      # (10 blocks over 3 ranks: nobody is without)
        frame = torch.full((H, W, 3), float(b1 - b0), dtype=torch.float32) + torch.arange(H * W * 3).reshape(H, W, 3) * (rank + 1)
        reduce_partial_sums(frame, dist)
        np.save(os.path.join(out_dir, f"sum{rank}.npy"), frame.numpy())
    finally:
        dist.destroy_process_group()


def test_sample_range_reduce(tmp_path):
    world = 3
    ranges = block_ranges(37, 4, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == 10 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    # a rank without a block gets None, never the (0, 0) that rta_render reads as "all blocks"
    assert block_ranges(8, 0, 2) == [None, (0, 1)] and block_ranges(8, 2, 2) == [(0, 2), (2, 4)]
    assert block_ranges(8, 4, 5) == [None, None, (0, 1), None, (1, 2)]
    assert all(r is None or r[0] < r[1] for w in range(1, 9) for r in block_ranges(20, 3, w))
    mp.spawn(_reduce_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    frames = [np.load(tmp_path / f"sum{r}.npy") for r in range(world)]
    want = 10.0 + np.arange(H * W * 3, dtype=np.float32).reshape(H, W, 3) * 6
    for f in frames:
        assert np.array_equal(f, want)
