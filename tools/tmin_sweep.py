#!/usr/bin/env python3
"""Segments per sample and image mean against tmin, f32 and f64 kernels (config 3 at 256 spp, BVH)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayz_amd import capi, render, tracer
render.init(0)
for prec, name, tmins in ((capi.PRECISION_F64, "f64", (1e-10, 1e-6, 1e-4, 1e-3)), (capi.PRECISION_F32, "f32", (1e-3, 3e-4, 1e-4, 3e-5, 1e-5, 1e-6))):
    for tmin in tmins:
        t = tracer.randomBouncing(1920, -50, 50, seed=42)
        t.samples_per_px = 256
        t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH, precision=prec, tmin=tmin)
        got, st = render.render_host(t.scene_desc(), t.camera_desc(), t.params())
        print(f"{name} tmin {tmin:7.0e}: segments/sample {st.segments / st.primary_rays:.5f}  image mean {got.astype(np.float64).mean():.6f}  "
              f"ground band mean {got[700:].astype(np.float64).mean():.6f}", flush=True)
