#!/usr/bin/env python3
"""Scheduling thresholds of the one-path BVH kernel on the SMALL scenes (config 2 = randomBouncing as shipped, 485 spheres;
a 4,097-sphere grid between it and config 3): rayz_hip_debug_set BVH_KEEP = keep_active | keep_stepping << 8."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer
render.init(0)
def bench(t, spp, reps=3):
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    out = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda")
    ds = render.DeviceScene(scene)
    st0 = torch.cuda.current_stream().cuda_stream
    ds.render_into(cam, p, out.data_ptr(), st0); ds.sync()
    best = 1e9
    for _ in range(reps):
        ds.render_into(cam, p, out.data_ptr(), st0); st = ds.sync(); best = min(best, st.kernel_ms)
    ds.close()
    return st.primary_rays / best / 1e3
c2 = tracer.randomBouncing(1920, seed=42)
c2b = tracer.randomBouncing(1920, -32, 32, seed=42)
pairs = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or [(24, 16), (32, 16), (40, 16), (48, 16), (32, 24), (40, 24), (40, 32), (48, 32), (56, 32), (16, 16), (24, 24)]
for ka, ks in pairs:
    render.debug_set(capi.DEBUG_BVH_KEEP, ka | (ks << 8))
    print(f"keep_active {ka:2d} keep_stepping {ks:2d}: config2 {bench(c2, 256):8.1f}   4,097 spheres {bench(c2b, 128):8.1f} Msamples/s", flush=True)
