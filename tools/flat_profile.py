#!/usr/bin/env python3
"""Per-phase wave time of the flat-list kernel (needs a -DRAYZ_FLAT_PROFILE build): scenes of several sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rayz_amd import capi, render, tracer
render.init(0)
for g in (5, 11, 50):
    t = tracer.randomBouncing(1920, -g, g, seed=42)
    t.samples_per_px = 64 if g < 50 else 16
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_LINEAR)
    print(f"grid {g}: {t.info().n_spheres} spheres", flush=True)
    got, st = render.render_host(t.scene_desc(), t.camera_desc(), t.params())
    print(f"   {st.primary_rays / st.kernel_ms / 1e3:.1f} Msamples/s", flush=True)
