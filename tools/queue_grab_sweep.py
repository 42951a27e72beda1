#!/usr/bin/env python3
"""Work items a wave reserves per atomic on the queue head (RAYZ_DEBUG_QUEUE_GRAB), re-swept in round 4 under the 64-sample chunk
schedule: BVH kernel on configs 3 / 2 / 5 (whole frame and one GPU's share of an 8-way deal of config 3), flat list on config 3.
    python tools/queue_grab_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer
from rayz_amd import dist as rdist

render.init(0)
st0 = torch.cuda.current_stream().cuda_stream
cases = [("config3 bvh 1024 spp", tracer.randomBouncing(1920, -50, 50, seed=42), 1024, capi.TRAVERSAL_BVH, 1),
         ("config3 bvh 1024 spp, shard 0 of 8", tracer.randomBouncing(1920, -50, 50, seed=42), 1024, capi.TRAVERSAL_BVH, 8),
         ("config2 bvh 256 spp", tracer.randomBouncing(1920, seed=42), 256, capi.TRAVERSAL_BVH, 1),
         ("config5 bvh 512 spp", tracer.triangleMesh(1920, 224, seed=1), 512, capi.TRAVERSAL_BVH, 1),
         ("config5 bvh 512 spp, shard 0 of 8", tracer.triangleMesh(1920, 224, seed=1), 512, capi.TRAVERSAL_BVH, 8),
         ("config3 flat 256 spp", tracer.randomBouncing(1920, -50, 50, seed=42), 256, capi.TRAVERSAL_LINEAR, 1),
         ("config3 flat 256 spp, shard 0 of 8", tracer.randomBouncing(1920, -50, 50, seed=42), 256, capi.TRAVERSAL_LINEAR, 8)]
for name, t, spp, trav, world in cases:
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=trav)
    ds = render.DeviceScene(t.scene_desc())
    cam = t.camera_desc()
    p = rdist.shard_params(t.params(), 0, world)
    out = torch.empty((render.shard_rows(p), p.width, 3), dtype=torch.float32, device="cuda")
    row = []
    for qg in (32, 64, 96, 128, 192, 256, 512):
        render.debug_set(capi.DEBUG_QUEUE_GRAB, qg)
        best = 1e9
        for _ in range(4):
            ds.render_into(cam, p, out.data_ptr(), st0)
            st = ds.sync()
            best = min(best, st.kernel_ms)
        row.append(f"{qg}: {st.primary_rays / best / 1e3:7.1f}")
    render.debug_set(capi.DEBUG_QUEUE_GRAB, -1)
    print(f"{name:38s} " + "  ".join(row), flush=True)
    ds.close()
