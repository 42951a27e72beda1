#!/usr/bin/env python3
"""Wavefront BVH traversal vs the persistent BVH kernel: bit-identical images, then timing on configs 3 / 5 / 2."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rayz_amd import capi, render, tracer

render.init(0)


def run(t, spp, trav, reps=1, w=None):
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=trav)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    out = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda")
    ds = render.DeviceScene(scene)
    st0 = torch.cuda.current_stream().cuda_stream
    ds.render_into(cam, p, out.data_ptr(), st0)
    st = ds.sync()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        ds.render_into(cam, p, out.data_ptr(), st0)
        st = ds.sync()
        best = min(best, time.perf_counter() - t0)
    img = out.cpu().numpy()
    ds.close()
    return img, st, best


for name, mk, spp in [("config3-small", lambda: tracer.randomBouncing(192, -50, 50, seed=42), 8),
                      ("config5-small", lambda: tracer.triangleMesh(192, 64, seed=1), 8),
                      ("config2-small", lambda: tracer.randomBouncing(160, seed=42), 16)]:
    a, sa, _ = run(mk(), spp, capi.TRAVERSAL_BVH)
    b, sb, _ = run(mk(), spp, capi.TRAVERSAL_BVH_WAVEFRONT)
    print(f"{name}: identical {np.array_equal(a, b)}  segments {sa.segments} {sb.segments}  box tests {sa.node_tests} {sb.node_tests}  "
          f"max|d| {np.abs(a - b).max():.3e}", flush=True)

spp3, spp5, spp2 = [int(x) for x in sys.argv[1:4]] + [256, 128, 64][len(sys.argv[1:4]):]
for name, mk, spp in [("config3", lambda: tracer.randomBouncing(1920, -50, 50, seed=42), spp3),
                      ("config5", lambda: tracer.triangleMesh(1920, 224, seed=1), spp5),
                      ("config2", lambda: tracer.randomBouncing(1920, seed=42), spp2)]:
    for trav, tn in ((capi.TRAVERSAL_BVH, "persistent"), (capi.TRAVERSAL_BVH_WAVEFRONT, "wavefront ")):
        img, st, dt = run(mk(), spp, trav, reps=2)
        print(f"{name} {tn}: {st.primary_rays / dt / 1e6:8.1f} Msamples/s  wall {dt * 1e3:8.2f} ms  kernel {st.kernel_ms:8.2f} ms  "
              f"{st.node_tests / st.segments:.1f} box/seg  mean {img.mean():.6f}", flush=True)
