#!/usr/bin/env python3
"""Quick timing of the f64 kernels (flat list and BVH) next to the f32 flat list, config-3 scene at reduced spp."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer
render.init(0)
def bench(name, t, spp, precision, traversal, reps=2):
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=traversal, precision=precision)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    dt = torch.float64 if precision == capi.PRECISION_F64 else torch.float32
    out = torch.empty((p.height, p.width, 3), dtype=dt, device="cuda")
    ds = render.DeviceScene(scene)
    st0 = torch.cuda.current_stream().cuda_stream
    run = lambda: ds.render_into(cam, p, out.data_ptr(), st0)
    run(); ds.sync()
    best = 1e9
    for _ in range(reps):
        run(); st = ds.sync(); best = min(best, st.kernel_ms)
    ds.close()
    print(f"{name}: {st.primary_rays / best / 1e3:8.1f} Msamples/s  kernel {best:8.2f} ms  mean {float(out.double().mean()):.9f}", flush=True)
t3 = tracer.randomBouncing(1920, -50, 50, seed=42)
bench("config3 f64 flat x16", t3, 16, capi.PRECISION_F64, capi.TRAVERSAL_LINEAR)
bench("config3 f64 bvh x64", t3, 64, capi.PRECISION_F64, capi.TRAVERSAL_BVH)
bench("config3 f32 flat x64", t3, 64, capi.PRECISION_F32, capi.TRAVERSAL_LINEAR)
bench("config2 f32 flat x256", tracer.randomBouncing(1920, seed=42), 256, capi.PRECISION_F32, capi.TRAVERSAL_LINEAR)
bench("config2 f64 flat x64", tracer.randomBouncing(1920, seed=42), 64, capi.PRECISION_F64, capi.TRAVERSAL_LINEAR)
