#!/usr/bin/env python3
"""What the row deal alone allows at N = 1, 2, 4, 8 GPUs (round 4; the development pool has ONE GPU per box, so no N > 1 frame has run on
N devices): every shard of an N-way deal of the BASELINE config-3 frame is rendered on this one GPU, one after the other.  The slowest
shard's kernel time is the frame time of an N-GPU node before the gather (one all_gather of 24.9 MB / N per rank over xGMI: an
estimated 0.1 ms); N x (1-GPU time) / that = the speed-up the deal permits.  python tools/predicted_scaling.py [spp_flat spp_bvh]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rayz_amd import capi, render, tracer
from rayz_amd import dist as rdist

render.init(0)
a = [int(x) for x in sys.argv[1:3]] + [1024, 1024][len(sys.argv[1:3]):]
t = tracer.randomBouncing(1920, -50, 50, seed=42)
ds = render.DeviceScene(t.scene_desc())
st0 = torch.cuda.current_stream().cuda_stream
for trav, spp, name, reps in ((capi.TRAVERSAL_LINEAR, a[0], "flat list (headline)", 1), (capi.TRAVERSAL_BVH, a[1], "BVH", 3)):
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=trav)
    cam, p0 = t.camera_desc(), t.params()
    one = None
    for world in (1, 2, 4, 8):
        ms = []
        for rank in range(world):
            p = rdist.shard_params(p0, rank, world)
            out = torch.empty((render.shard_rows(p), p.width, 3), dtype=torch.float32, device="cuda")
            if world == 1 and rank == 0:  # (first launch: uploads)
                ds.render_into(cam, p, out.data_ptr(), st0)
                ds.sync()
            best = 1e9
            for _ in range(reps):
                ds.render_into(cam, p, out.data_ptr(), st0)
                best = min(best, ds.sync().kernel_ms)
            ms.append(best)
        k = np.array(ms)
        one = one or k.max()
        total = p0.width * p0.height * spp
        print(f"{name} {spp} spp, N = {world}: shard kernel ms min {k.min():9.2f} mean {k.mean():9.2f} max {k.max():9.2f} -> {total / k.max() / 1e3:9.1f} Msamples/s, "
              f"speed-up {one / k.max():5.2f}x = {100 * one / k.max() / world:5.1f} % of linear (work / N: {100 * one / world / k.mean():5.1f} % — the rest is imbalance between shards)", flush=True)
ds.close()
