#!/usr/bin/env python3
"""Sweep the one-path BVH kernel's scheduling thresholds (rayz_hip_debug_set BVH_KEEP = active | stepping << 8) on config 3 / config 5."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer

render.init(0)
def bench(t, spp, reps=2):
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
    return st.primary_rays / best / 1e3, st.node_tests / st.segments

c3 = tracer.randomBouncing(1920, -50, 50, seed=42)
c5 = tracer.triangleMesh(1920, 224, seed=1)
pairs = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or [(40, 24), (48, 32), (56, 40), (32, 16), (24, 12), (16, 8), (48, 16), (56, 8), (32, 32), (60, 48), (1, 1)]
for ka, ks in pairs:
    render.debug_set(capi.DEBUG_BVH_KERNEL, 1)
    render.debug_set(capi.DEBUG_BVH_KEEP, ka | (ks << 8))
    a, na = bench(c3, 256)
    b, nb = bench(c5, 128)
    print(f"keep_active {ka:2d} keep_stepping {ks:2d}: config3 {a:8.1f} Msamples/s ({na:.1f} nodes/seg)   config5 {b:8.1f} ({nb:.1f})", flush=True)
