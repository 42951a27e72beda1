#!/usr/bin/env python3
"""config 3 / 5 / 2 through the BVH kernel at the library's default schedule, and the top-of-tree size (debug BVH_TOP)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer
render.init(0)
def bench(name, t, spp, reps=3):
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
for top in [int(x) for x in sys.argv[1:]] or [256]:
    render.debug_set(capi.DEBUG_BVH_TOP, top)
    print(f"top {top:4d}: config3 {bench('c3', tracer.randomBouncing(1920, -50, 50, seed=42), 256):8.1f}  config5 {bench('c5', tracer.triangleMesh(1920, 224, seed=1), 128):8.1f}  "
          f"config2 {bench('c2', tracer.randomBouncing(1920, seed=42), 64):8.1f} Msamples/s", flush=True)
