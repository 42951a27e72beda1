#!/usr/bin/env python3
"""Flat list vs BVH by scene size (where RAYZ_AUTO_BVH_MIN should sit): randomBouncing grids at 1920x1080x64spp."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer

render.init(0)
for g in (3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 20, 32):
    row = []
    for trav in (capi.TRAVERSAL_LINEAR, capi.TRAVERSAL_BVH):
        t = tracer.randomBouncing(1920, -g, g, seed=42)
        t.samples_per_px = int(os.environ.get("RAYZ_CROSSOVER_SPP", "64"))
        t.set_gpu(render_seed=1, traversal=trav)
        scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
        out = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda")
        ds = render.DeviceScene(scene)
        st0 = torch.cuda.current_stream().cuda_stream
        ds.render_into(cam, p, out.data_ptr(), st0); ds.sync()
        best = 1e9
        for _ in range(2):
            ds.render_into(cam, p, out.data_ptr(), st0); st = ds.sync(); best = min(best, st.kernel_ms)
        ds.close()
        row.append(st.primary_rays / best / 1e3)
    print(f"grid {g:2d}: {t.info().n_spheres:5d} spheres   flat {row[0]:8.1f}   bvh {row[1]:8.1f} Msamples/s   bvh/flat {row[1] / row[0]:.2f}", flush=True)
