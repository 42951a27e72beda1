#!/usr/bin/env python3
"""Largest chunk of the automatic (graded) schedule vs whole-frame rate and the rate of ONE GPU's share of an 8-way deal, for the
BASELINE configs at their own sizes (round 4).  RAYZ_DEBUG_CHUNK_CAP changes the image and exists in -DRAYZ_EXPERIMENTS builds only:
    bash tools/build_experiments.sh && bash tools/with_lib.sh variants/lib_experiments_s4.so python tools/chunk_cap_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer
from rayz_amd import dist as rdist

render.init(0)
st0 = torch.cuda.current_stream().cuda_stream
cases = [("config2 bvh 1080p x 256", tracer.randomBouncing(1920, seed=42), 256, capi.TRAVERSAL_BVH),
         ("config2 flat 1080p x 256", tracer.randomBouncing(1920, seed=42), 256, capi.TRAVERSAL_LINEAR),
         ("config5 bvh 1080p x 512", tracer.triangleMesh(1920, 224, seed=1), 512, capi.TRAVERSAL_BVH),
         ("config3 bvh 1080p x 1024", tracer.randomBouncing(1920, -50, 50, seed=42), 1024, capi.TRAVERSAL_BVH),
         ("config4 bvh 4K x 1024 (of 4096)", tracer.randomBouncing(3840, -50, 50, seed=42), 1024, capi.TRAVERSAL_BVH)]
for name, t, spp, trav in cases:
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=trav)
    ds = render.DeviceScene(t.scene_desc())
    cam, p0 = t.camera_desc(), t.params()
    for cap in (256, 128, 64, 32, 16):
        render.debug_set(capi.DEBUG_CHUNK_CAP, cap)
        res = []
        for world in (1, 8):
            p = rdist.shard_params(p0, 0, world)
            rows = render.shard_rows(p)
            out = torch.empty((rows, p.width, 3), dtype=torch.float32, device="cuda")
            best = 1e9
            for _ in range(4):
                ds.render_into(cam, p, out.data_ptr(), st0)
                st = ds.sync()
                best = min(best, st.kernel_ms)
            res.append(rows * p.width * spp / best / 1e3)
        print(f"{name:34s} cap {cap:3d}: whole frame {res[0]:8.1f} Msamples/s | shard 0 of 8 {res[1]:8.1f} = {100 * res[1] / res[0]:5.1f} %", flush=True)
    render.debug_set(capi.DEBUG_CHUNK_CAP, -1)
    ds.close()
