#!/usr/bin/env python3
"""tile_rows A/B at BASELINE config 3's FULL size (1024 spp), flat list and BVH: every shard of an 8-way deal on this one GPU, slowest
shard = the frame time of an 8-GPU node before the gather (round 4; the 256-spp version is tools/tile_rows_ab.py).
    python tools/tile_rows_ab_full.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rayz_amd import capi, render, tracer
from rayz_amd import dist as rdist
render.init(0)
t = tracer.randomBouncing(1920, -50, 50, seed=42)
ds = render.DeviceScene(t.scene_desc())
st0 = torch.cuda.current_stream().cuda_stream
for trav, name, reps in ((capi.TRAVERSAL_LINEAR, "flat", 1), (capi.TRAVERSAL_BVH, "bvh", 3)):
    t.samples_per_px = 1024
    t.set_gpu(render_seed=1, traversal=trav)
    cam, p0 = t.camera_desc(), t.params()
    for tr in (1, 2, 4, 8):
        ms = []
        for rank in range(8):
            p = rdist.shard_params(p0, rank, 8, tile_rows=tr)
            out = torch.empty((render.shard_rows(p), p.width, 3), dtype=torch.float32, device="cuda")
            best = 1e9
            for _ in range(reps):
                ds.render_into(cam, p, out.data_ptr(), st0); best = min(best, ds.sync().kernel_ms)
            ms.append(best)
        k = np.array(ms)
        print(f"{name} 1024 spp, 8 shards, tile_rows {tr}: min {k.min():8.1f} mean {k.mean():8.1f} max {k.max():8.1f} ms", flush=True)
