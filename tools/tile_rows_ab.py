#!/usr/bin/env python3
"""tile_rows A/B for the multi-GPU row deal (round 4): the BVH kernel's 8x8 work tiles are built from a shard's LOCAL rows, so with
1-row interleave on 8 shards a tile spans 57 global rows; 8-row tiles keep them coherent but deal 1080 / 8 = 135 tiles unevenly
(17 vs 16 per rank).  Every shard of an 8-way frame is rendered on this one GPU, one after the other: max over shards = the frame
time of an 8-GPU node (gather apart), mean = the work.   python tools/tile_rows_ab.py [spp3 spp4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rayz_amd import capi, render, tracer
from rayz_amd import dist as rdist

render.init(0)
a = [int(x) for x in sys.argv[1:3]] + [256, 256][len(sys.argv[1:3]):]


def run(name, t, spp, world=8):
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
    scene, cam, p0 = t.scene_desc(), t.camera_desc(), t.params()
    ds = render.DeviceScene(scene)
    st0 = torch.cuda.current_stream().cuda_stream
    for tr in (1, 2, 4, 8, 16):
        ms = []
        for rank in range(world):
            p = rdist.shard_params(p0, rank, world, tile_rows=tr)
            rows = render.shard_rows(p)
            out = torch.empty((rows, p.width, 3), dtype=torch.float32, device="cuda")
            best = 1e9
            for _ in range(3):
                ds.render_into(cam, p, out.data_ptr(), st0)
                st = ds.sync()
                best = min(best, st.kernel_ms)
            ms.append((best, rows))
        k = np.array([m for m, _ in ms])
        total = p0.width * p0.height * spp
        print(f"{name} tile_rows {tr:2d}: shard kernel ms min {k.min():8.2f} mean {k.mean():8.2f} max {k.max():8.2f}  rows per shard {min(r for _, r in ms)}..{max(r for _, r in ms)}  "
              f"8-GPU frame rate (max over shards) {total / k.max() / 1e3:9.1f} Msamples/s, work rate {total / k.sum() / 1e3:8.1f} per GPU", flush=True)
    ds.close()


run("config3 1920x1080", tracer.randomBouncing(1920, -50, 50, seed=42), a[0])
run("config4 3840x2160", tracer.randomBouncing(3840, -50, 50, seed=42), a[1])
