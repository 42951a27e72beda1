#!/usr/bin/env python3
"""Quick timing of the flat-list kernel: configs 1, 2 (full), 3 (reduced spp), a 100-sphere scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer
render.init(0)
def bench(name, t, spp, reps=3):
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_LINEAR)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    out = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda")
    ds = render.DeviceScene(scene)
    st0 = torch.cuda.current_stream().cuda_stream
    ds.render_into(cam, p, out.data_ptr(), st0); ds.sync()
    best = 1e9
    for _ in range(reps):
        ds.render_into(cam, p, out.data_ptr(), st0); st = ds.sync(); best = min(best, st.kernel_ms)
    ds.close()
    print(f"{name}: {st.primary_rays / best / 1e3:8.1f} Msamples/s  kernel {best:8.2f} ms  mean {float(out.mean()):.6f}", flush=True)
bench("config1 3 spheres 400x225x8", tracer.threeSpheres(400, seed=1), 8)
bench("3 spheres 1920x1080x64", tracer.threeSpheres(1920, seed=1), 64)
bench("100 spheres (grid 5) 1920x1080x64", tracer.randomBouncing(1920, -5, 5, seed=42), 64)
bench("config2 485 spheres 1920x1080x256", tracer.randomBouncing(1920, seed=42), 256)
bench("config3 10k spheres 1920x1080x64", tracer.randomBouncing(1920, -50, 50, seed=42), 64)
