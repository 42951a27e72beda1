#!/usr/bin/env python3
"""Which inner nodes the LDS top holds: grown from the root by box surface area (default) vs breadth-first
(rayz_hip_debug_set BVH_TOP_ORDER); configs 3 / 5 / 2 through the BVH, alternating."""
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
    return st.primary_rays / best / 1e3, float(out.double().sum())
c3, c5, c2 = tracer.randomBouncing(1920, -50, 50, seed=42), tracer.triangleMesh(1920, 224, seed=1), tracer.randomBouncing(1920, seed=42)
sums = {}
for rep in range(2):
    for order, name in ((0, "surface area"), (1, "breadth-first")):
        render.debug_set(capi.DEBUG_BVH_TOP_ORDER, order)
        r = [bench(c3, 256), bench(c5, 128), bench(c2, 256)]
        sums.setdefault(order, [x[1] for x in r])
        print(f"{name:14s}: config3 {r[0][0]:8.1f}  config5 {r[1][0]:8.1f}  config2 {r[2][0]:8.1f} Msamples/s", flush=True)
assert sums[0] == sums[1], "the order of the top changed an image"
render.debug_set(capi.DEBUG_BVH_TOP_ORDER, -1)
