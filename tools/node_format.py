#!/usr/bin/env python3
"""BVH node record format (debug BVH_NODES: 1 = f32 planes, 64 B; 2 = 16-bit plane indices, 32 B) against tree size:
randomBouncing grids of growing extent (n ≈ (2·g)² spheres), config 2/3 and the config-5 mesh.  Same image either way."""
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
    return st.primary_rays / best / 1e3, out.clone()
scenes = [("config2 g11", lambda: tracer.randomBouncing(1920, seed=42), 64)]
for g in (25, 50, 75, 100, 150, 220):
    scenes.append((f"grid {g} ({(2*g)**2} spheres)" + (" = config3" if g == 50 else ""), (lambda g=g: tracer.randomBouncing(1920, -g, g, seed=42)), 128))
scenes.append(("config5 mesh", lambda: tracer.triangleMesh(1920, 224, seed=1), 128))
for name, make, spp in scenes:
    res = {}
    for fmt in (1, 2, 0):
        render.debug_set(capi.DEBUG_BVH_NODES, fmt)
        res[fmt] = bench(make(), spp)
    same = torch.equal(res[1][1], res[2][1]) and torch.equal(res[0][1], res[1][1])
    print(f"{name:32s} f32 planes {res[1][0]:8.1f}  16-bit {res[2][0]:8.1f}  ({res[2][0]/res[1][0]-1:+.1%})  auto {res[0][0]:8.1f}  same image {same}", flush=True)
