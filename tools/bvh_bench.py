#!/usr/bin/env python3
"""Quick timing of the BVH kernel on configs 3, 5 and 2 (reduced spp; kernel time from the library's HIP events).
    python tools/bvh_bench.py [spp3 spp5 spp2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer

render.init(0)
if os.environ.get("RAYZ_BENCH_QUEUE_GRAB"):
    render.debug_set(capi.DEBUG_QUEUE_GRAB, int(os.environ["RAYZ_BENCH_QUEUE_GRAB"]))  # (experiments: items reserved per queue atomic)


def bench(name, t, spp, reps=3, trav=capi.TRAVERSAL_BVH):
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=trav)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    p.chunk_spp = int(os.environ.get("RAYZ_BENCH_CHUNK_SPP", "0"))  # (uniform chunks instead of the automatic schedule: per-item cost experiments)
    out = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda")
    ds = render.DeviceScene(scene)
    st0 = torch.cuda.current_stream().cuda_stream
    ds.render_into(cam, p, out.data_ptr(), st0)
    ds.sync()
    best = 1e9
    for _ in range(reps):
        ds.render_into(cam, p, out.data_ptr(), st0)
        st = ds.sync()
        best = min(best, st.kernel_ms)
    ds.close()
    print(f"{name}: {st.primary_rays / best / 1e3:8.1f} Msamples/s  kernel {best:8.2f} ms  {st.node_tests / st.segments:.1f} box tests/seg "
          f"{st.sphere_tests / st.segments:.2f} leaf tests/seg  {st.segments / st.primary_rays:.3f} seg/sample  mean {float(out.mean()):.6f}", flush=True)


a = [int(x) for x in sys.argv[1:4]] + [256, 128, 64][len(sys.argv[1:4]):]
bench("config3 bvh", tracer.randomBouncing(1920, -50, 50, seed=42), a[0])
bench("config5 bvh", tracer.triangleMesh(1920, 224, seed=1), a[1])
bench("config2 bvh", tracer.randomBouncing(1920, seed=42), a[2])
bench("config2 flat", tracer.randomBouncing(1920, seed=42), a[2], trav=capi.TRAVERSAL_LINEAR)
