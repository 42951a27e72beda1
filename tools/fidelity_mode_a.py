#!/usr/bin/env python3
"""Full-frame fidelity of the GPU kernels against the reference as written (oracle mode A: f64, its BVH, sequential
xoshiro streams, tmin 1e-10) on BASELINE config 2 (randomBouncing, 1920x1080).  Mode A renders SPP_A samples per
pixel on all host cores (with per-pixel sample variances); the GPU renders 4096 spp in f64 (tmin 1e-10, the
reference's arithmetic) and in f32 (the default).  Reported: image and row-band means against A's standard errors
and the distribution of per-pixel z-scores."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import binding as oracle
from rayz_amd import capi, render, tracer

SPP_A = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
render.init(0)
oracle.load()
t = tracer.randomBouncing(W, seed=42)
scene, cam = t.scene_desc(), t.camera_desc()
pa = t.params()
pa.precision, pa.tmin, pa.samples_per_px = capi.PRECISION_F64, 1e-10, SPP_A
H = pa.height
threads = max(1, min(os.cpu_count() or 1, 32))
a = np.zeros((H, W, 3)); sq = np.zeros((H, W, 3))
def work(k):
    st = t.rng_state().copy(); st[0] ^= (0x9E3779B97F4A7C15 * (k + 1)) & 0xFFFFFFFFFFFFFFFF
    for r in range(k, H, threads):
        img, s2, _ = oracle.render_a(scene, cam, pa, st, row_begin=r, row_end=r + 1, want_sumsq=True)
        a[r], sq[r] = img[0], s2[0]
t0 = time.time()
with ThreadPoolExecutor(threads) as ex:
    list(ex.map(work, range(threads)))
print(f"mode A: {W}x{H} at {SPP_A} spp on {threads} threads in {time.time() - t0:.1f} s", flush=True)
var = np.maximum(sq / SPP_A - a ** 2, 0) / SPP_A  # variance of A's pixel means
for name, prec, tmin in (("GPU f64 (tmin 1e-10)", capi.PRECISION_F64, 1e-10), ("GPU f32 (tmin 1e-3)", capi.PRECISION_F32, 1e-3)):
    t.samples_per_px = 4096
    t.set_gpu(render_seed=5, precision=prec, tmin=tmin, traversal=capi.TRAVERSAL_LINEAR)
    g, st = render.render_host(scene, cam, t.params())
    g = g.astype(np.float64)
    se = np.sqrt(var.sum()) / a.size
    z = (a - g) / np.sqrt(np.maximum(var, 1e-14))
    ok = var > 1e-12
    bands = np.array_split(np.arange(H), 12)
    bz = [(a[b].mean() - g[b].mean()) / (np.sqrt(var[b].sum()) / a[b].size) for b in bands]
    print(f"{name}: image mean A {a.mean():.6f} GPU {g.mean():.6f}  diff {(a.mean() - g.mean()) / se:+.2f} SE ({(a.mean() / g.mean() - 1):+.2e} rel)")
    print(f"    row-band z: " + " ".join(f"{v:+.1f}" for v in bz))
    print(f"    per-pixel z: median {np.median(z[ok]):+.3f}, std {z[ok].std():.3f}, |z|>4: {(np.abs(z[ok]) > 4).mean():.2e}, segments/sample {st.segments / st.primary_rays:.4f}", flush=True)
