#!/usr/bin/env python3
"""Diagnostic: render big sample counts and report non-finite / negative pixels (rare-event hunt).
usage: nan_hunt.py <scene> <width> <spp> [bvh|linear] [f32|f64] [bounces]
scenes: bouncing10k, bouncing, three, mesh, custom (tests' scene with every material kind and diffuse method)"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rayz_amd import capi, render, tracer

render.init(0)
name, w, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
trav = sys.argv[4] if len(sys.argv) > 4 else "bvh"
prec = sys.argv[5] if len(sys.argv) > 5 else "f32"
if name == "bouncing10k":
    t = tracer.randomBouncing(w, -50, 50, seed=42)
elif name == "bouncing":
    t = tracer.randomBouncing(w, seed=7)
elif name == "three":
    t = tracer.threeSpheres(w, seed=3)
elif name == "mesh":
    t = tracer.triangleMesh(w, 224, seed=1)
else:
    from test_gpu_parity import _custom_scene
    t = _custom_scene()
t.samples_per_px = spp
if len(sys.argv) > 6:
    t.max_bounces = int(sys.argv[6])
t.set_gpu(render_seed=int(os.environ.get("RAYZ_HUNT_SEED", "1")), traversal=capi.TRAVERSAL_BVH if trav == "bvh" else capi.TRAVERSAL_LINEAR,
          precision=capi.PRECISION_F64 if prec == "f64" else capi.PRECISION_F32)
scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
t0 = time.time()
got, st = render.render_host(scene, cam, p)
bad = ~np.isfinite(got).all(axis=2) | (got < 0).any(axis=2)
idx = np.flatnonzero(bad.reshape(-1))
print(f"{name} {p.width}x{p.height} {spp} spp {trav} {prec}: {idx.size} bad pixels of {bad.size}; {st.primary_rays:.3g} samples, "
      f"segments/sample {st.segments / st.primary_rays:.3f}, max {np.nanmax(got):.3f}, {time.time() - t0:.1f} s", flush=True)
for i in idx[:8]:
    print("   ", int(i), int(i) // p.width, int(i) % p.width, got.reshape(-1, 3)[i])
