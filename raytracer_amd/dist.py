"""Assembling one frame from per-GPU shards with torch.distributed.

Two ways to split a frame (raytracer_amd.h, rta_render_params):

* tiles (ShardGather, the default): exact, one all-gather;
* sample ranges (block_ranges + reduce_partial_sums): every rank renders all
  pixels for its share of the sample blocks, one all-reduce(sum) of the frame;
  equal to the unsplit frame up to the order of the last fp32 additions.

Each rank renders the 8x8 tiles with tile_id % world == rank
(rta_shard_pixels); the radiance sums of all ranks are brought together with
ONE all-gather of equally sized (padded) shard buffers - over RCCL/xGMI when
the tensors live on GPUs, over gloo in the CPU tests - and scattered back to
pixel positions.  Nothing else is exchanged: pixels are independent.
"""
import numpy as np
import torch

from . import shard_pixels


class ShardGather:
    def __init__(self, width, height, rank, world, device):
        self.rank, self.world = rank, world
        lists = [shard_pixels(width, height, r, world).astype(np.int64) for r in range(world)]
        self.counts = [len(x) for x in lists]
        self.pad = max(self.counts) if self.counts else 0
        self.mine = torch.from_numpy(lists[rank]).to(device)
        self.everyone = torch.from_numpy(np.concatenate(lists)).to(device)
        self.send = torch.zeros((self.pad, 3), dtype=torch.float32, device=device)
        self.recv = torch.zeros((world, self.pad, 3), dtype=torch.float32, device=device)

    def gather(self, frame, dist):
        """frame: [height, width, 3] float32 holding this rank's pixels; on
        return it holds every rank's pixels."""
        if self.world == 1:
            return frame
        flat = frame.view(-1, 3)
        self.send[: self.counts[self.rank]] = flat[self.mine]
        if self.send.is_cuda and dist.get_backend() == "gloo":
            # gloo gathers host tensors only (rehearsals of the N > 1 path on one GPU)
            recv = torch.empty(self.recv.shape, dtype=self.recv.dtype)
            dist.all_gather_into_tensor(recv.view(-1, 3), self.send.cpu())
            self.recv.copy_(recv)
        else:
            dist.all_gather_into_tensor(self.recv.view(-1, 3), self.send)
        parts = [self.recv[r, : self.counts[r]] for r in range(self.world)]
        flat[self.everyone] = torch.cat(parts)
        return frame

    @property
    def bytes_per_rank(self):
        return self.pad * 12


def block_ranges(samples, samples_per_stream, world):
    """The sample-block share of every rank: the blocks cut into `world` near-equal runs, each a
    (begin, end) pair with begin < end - or None for a rank that gets no block (more ranks than
    blocks).  A None share must not reach rta_render: there, block_begin == block_end == 0 asks for
    ALL blocks.  render_share() below does the right thing for either."""
    # (samples_per_stream <= 0 means one stream, hence one block, per pixel: rta_render_params)
    n_blocks = 1 if samples_per_stream <= 0 else -(-samples // min(samples_per_stream, samples))
    cuts = [n_blocks * r // world for r in range(world + 1)]
    return [(a, b) if b > a else None for a, b in zip(cuts[:-1], cuts[1:])]


def render_share(ctx, camera, params, share, frame):
    """Renders one rank's share of the sample blocks (an entry of block_ranges) into `frame`, a
    [height, width, 3] float32 device tensor that receives the partial radiance sums; a rank
    without a share contributes zeros and renders nothing.  Returns the render's statistics."""
    from . import Stats
    if share is None:
        frame.zero_()
        if frame.is_cuda:
            import torch
            torch.cuda.current_stream(frame.device).synchronize()
        return dict.fromkeys(Stats().as_dict(), 0)
    params.block_begin, params.block_end = share
    return ctx.render_device(camera, params, frame.data_ptr(), None)


def reduce_partial_sums(frame, dist):
    """frame: [height, width, 3] float32 holding this rank's partial radiance sums over its block
    range; on return every rank holds the sums over all samples."""
    if dist is not None and dist.get_world_size() > 1:
        if frame.is_cuda and dist.get_backend() == "gloo":
            host = frame.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            frame.copy_(host)
        else:
            dist.all_reduce(frame, op=dist.ReduceOp.SUM)
    return frame
