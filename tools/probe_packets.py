#!/usr/bin/env python3
"""What do packets of 64 rays cost on the coherent soup batch when a packet is an 8x8 pixel tile instead of a
64x1 strip of a row?   RTA_BATCH_PACKETS=2 python tools/probe_packets.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracer_amd as rta
from raytracer_amd import synthetic

dev = torch.device("cuda", 0)
ctx = rta.Context(0)
scene = rta.Scene.from_triangles(synthetic.make_soup(10_000_000))
scene.build_bvh_on(ctx)
ctx.upload(scene)
side = 8192
rays = synthetic.coherent_rays(side, dev)
for tile_w, tile_h in ((64, 1), (8, 8), (16, 4), (4, 16)):
    r = rays.view(side // tile_h, tile_h, side // tile_w, tile_w, 6).permute(0, 2, 1, 3, 4).reshape(-1, 6).contiguous()
    hits = torch.empty((r.shape[0], 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for rep in range(3):
        ctx.trace_closest_device(r.data_ptr(), r.shape[0], hits.data_ptr())
    st = ctx.last_batch_stats()
    ctx.trace_closest_device(r.data_ptr(), r.shape[0], hits.data_ptr(), count_work=True)
    cw = ctx.last_batch_stats()
    print(f"packet {tile_w}x{tile_h}: packets {st['packets']}, walk {st['walk_ms']:.2f} ms, order {st['order_ms']:.2f}, replay {st['replay_ms']:.2f}, "
          f"records per packet: {cw['walk_node_visits'] * 64 / r.shape[0]:.1f} nodes, {cw['walk_tri_tests'] * 64 / r.shape[0]:.1f} triangles", flush=True)
