#!/usr/bin/env python3
"""Time each BASELINE.json config once on one GPU (device-resident, HIP-event kernel time + wall).
Prints one JSON object per line; used for DESIGN.md's table (profiles/rNN/configs.jsonl).

    python tools/measure_configs.py [--quick]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from rayz_amd import capi, render, tracer  # noqa: E402
from rayz_amd import dist as rdist  # noqa: E402


def run(name, t, spp, trav=capi.TRAVERSAL_LINEAR, prec=capi.PRECISION_F32, shard=(0, 1), reps=1):
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=trav, precision=prec)
    scene, cam = t.scene_desc(), t.camera_desc()
    p = rdist.shard_params(t.params(), shard[0], shard[1])
    rows = render.shard_rows(p)
    out = torch.empty((rows, p.width, 3), dtype=torch.float64 if prec else torch.float32, device="cuda")
    ds = render.DeviceScene(scene)
    stream = torch.cuda.current_stream().cuda_stream
    ds.render_into(cam, p, out.data_ptr(), stream)  # warm-up: uploads, workspace
    ds.sync()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        ds.render_into(cam, p, out.data_ptr(), stream)
        st = ds.sync()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, st)
    dt, st = best
    i = t.info()
    rec = {"config": name, "spheres": i.n_spheres, "triangles": i.n_triangles, "width": p.width, "rows": rows, "spp": spp,
           "traversal": "bvh" if trav else "linear", "precision": "f64" if prec else "f32", "shard": list(shard),
           "samples": st.primary_rays, "wall_s": dt, "kernel_ms": st.kernel_ms, "Msamples_per_s": st.primary_rays / dt / 1e6,
           "segments_per_sample": st.segments / st.primary_rays, "prim_tests_per_segment": st.sphere_tests / max(st.segments, 1),
           "node_tests_per_segment": st.node_tests / max(st.segments, 1), "mean_radiance": float(out.mean())}
    print(json.dumps(rec), flush=True)
    ds.close()


def main():
    quick = "--quick" in sys.argv
    render.init(0)
    L, B = capi.TRAVERSAL_LINEAR, capi.TRAVERSAL_BVH
    run("1: 3 Lambertian spheres 400x225x8", tracer.threeSpheres(400, seed=1), 8, L)
    run("2: RTIOW cover (randomBouncing) 1920x1080x256", tracer.randomBouncing(1920, seed=42), 256, L)
    run("2: same, BVH", tracer.randomBouncing(1920, seed=42), 256, B)
    c3 = lambda: tracer.randomBouncing(1920, -50, 50, seed=42)  # noqa: E731
    run("3: 10k spheres 1920x1080x1024 flat list", c3(), 64 if quick else 1024, L)
    run("3: same, BVH", c3(), 64 if quick else 1024, B)
    run("3: same, f64 fidelity mode, flat list (64 spp)", c3(), 64, L, capi.PRECISION_F64)
    run("3: same, f64 fidelity mode, BVH (256 spp)", c3(), 256, B, capi.PRECISION_F64)
    c4 = lambda: tracer.randomBouncing(3840, -50, 50, seed=42)  # noqa: E731
    run("4: 10k spheres 3840x2160x4096, shard 0 of 8 (one GPU's share), flat list", c4(), 64 if quick else 4096, L, shard=(0, 8))
    run("4: same shard, BVH", c4(), 64 if quick else 4096, B, shard=(0, 8))
    mesh = lambda: tracer.triangleMesh(1920, 224, seed=1)  # noqa: E731
    run("5: 100k-triangle mesh 1920x1080x512, BVH", mesh(), 32 if quick else 512, B)
    run("5: same, flat list (8 spp)", mesh(), 8, L)


if __name__ == "__main__":
    main()
