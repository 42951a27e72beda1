#!/usr/bin/env python3
"""How the walked tree is split (debug BVH_SPLIT: 0 = surface-area heuristic, 1 = the reference's median split):
Msamples/s and box / leaf tests per segment, configs 2, 3, 5.  Same image either way."""
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
    return st.primary_rays / best / 1e3, st.node_tests / st.segments, st.sphere_tests / st.segments, out.clone()
scenes = [("config2", lambda: tracer.randomBouncing(1920, seed=42), 128), ("config3", lambda: tracer.randomBouncing(1920, -50, 50, seed=42), 256),
          ("grid 150", lambda: tracer.randomBouncing(1920, -150, 150, seed=42), 128), ("config5 mesh", lambda: tracer.triangleMesh(1920, 224, seed=1), 128)]
for name, make, spp in (scenes[:int(sys.argv[1])] if len(sys.argv) > 1 else scenes):
    res = {}
    for split in (1, 0):
        render.debug_set(capi.DEBUG_BVH_SPLIT, split)
        res[split] = bench(make(), spp)
    m, s = res[1], res[0]
    print(f"{name:14s} median {m[0]:8.1f} Msamples/s ({m[1]:.1f} box, {m[2]:.2f} leaf tests/seg)   sah {s[0]:8.1f} ({s[1]:.1f}, {s[2]:.2f})   "
          f"{s[0]/m[0]-1:+.1%}   same image {torch.equal(m[3], s[3])}", flush=True)
