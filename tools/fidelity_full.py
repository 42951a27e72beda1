#!/usr/bin/env python3
"""Full-size fidelity check: config 3 (10,003 spheres, 1920x1080, 1024 spp) through the BVH kernel in f32 (f32 reject
test + f64 roots, tmin 1e-3) and in f64 (tmin 1e-10), different seeds; compares image means and segments per sample."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayz_amd import capi, render, tracer
render.init(0)
res = {}
for name, prec, seed in (("f32 seed 1", capi.PRECISION_F32, 1), ("f32 seed 2", capi.PRECISION_F32, 2), ("f64 seed 3", capi.PRECISION_F64, 3)):
    t = tracer.randomBouncing(1920, -50, 50, seed=42)
    t.samples_per_px = 1024
    t.set_gpu(render_seed=seed, traversal=capi.TRAVERSAL_BVH, precision=prec)
    got, st = render.render_host(t.scene_desc(), t.camera_desc(), t.params())
    got = got.astype(np.float64)
    res[name] = (got, st.segments / st.primary_rays)
    print(f"{name}: mean RGB {got.mean(axis=(0, 1))}, segments/sample {st.segments / st.primary_rays:.5f}", flush=True)
a, b, c = res["f32 seed 1"][0], res["f32 seed 2"][0], res["f64 seed 3"][0]
noise = np.abs(a - b).mean()
print(f"mean |f32(seed1) - f32(seed2)| per channel value: {noise:.3e}  (Monte-Carlo noise floor between two seeds)")
print(f"mean |f32(seed1) - f64(seed3)|                  : {np.abs(a - c).mean():.3e}")
print(f"image-mean difference f32 vs f32: {(a.mean() - b.mean()) / a.mean():+.2e} relative; f32 vs f64: {(a.mean() - c.mean()) / a.mean():+.2e} relative")
band = lambda x: x.reshape(20, 54, 1920, 3).mean(axis=(1, 2, 3))
print("max relative row-band (54 rows) mean difference f32 vs f32:", f"{np.abs(band(a) / band(b) - 1).max():.2e}", " f32 vs f64:", f"{np.abs(band(a) / band(c) - 1).max():.2e}")
