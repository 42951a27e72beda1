#!/usr/bin/env python3
"""Diagnostic: render a big frame through the BVH kernel and report non-finite / negative pixels."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayz_amd import capi, render, tracer

render.init(0)
w, spp = int(sys.argv[1]), int(sys.argv[2])
t = tracer.randomBouncing(w, -50, 50, seed=42)
t.samples_per_px = spp
t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
got, st = render.render_host(scene, cam, p)
bad = ~np.isfinite(got).all(axis=2) | (got < 0).any(axis=2)
idx = np.flatnonzero(bad.reshape(-1))
print(f"{w}x{p.height} {spp} spp: {idx.size} bad pixels of {bad.size}; segments/sample {st.segments / st.primary_rays:.3f}")
for i in idx[:20]:
    print(int(i), int(i) // w, int(i) % w, got.reshape(-1, 3)[i])
json.dump([int(i) for i in idx[:200]], open(os.path.join(ROOT, "gpurun_out", f"bad_{w}_{spp}.json"), "w"))
