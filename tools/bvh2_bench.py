#!/usr/bin/env python3
"""One-path vs two-path BVH kernel (rayz_hip_debug_set BVH_KERNEL) and a sweep of the two-path kernel's scheduling
thresholds (BVH2_KEEP = service | blocked << 8 | swap << 16 | keep_stepping << 24) on configs 3 / 5 / 2.
    bash tools/build_experiments.sh && bash tools/with_lib.sh variants/lib_experiments_s4.so python tools/bvh2_bench.py [--sweep] [spp3 spp5 spp2]
(the two-path kernel is not in the product library: -DRAYZ_EXPERIMENTS)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer

render.init(0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
spp = [int(x) for x in args[:3]] + [256, 128, 64][len(args[:3]):]
scenes = {"config3": (tracer.randomBouncing(1920, -50, 50, seed=42), spp[0]), "config5": (tracer.triangleMesh(1920, 224, seed=1), spp[1]),
          "config2": (tracer.randomBouncing(1920, seed=42), spp[2])}
ds_cache = {}


def bench(name, reps=3):
    t, n = scenes[name]
    t.samples_per_px = n
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    if name not in ds_cache:
        ds_cache[name] = (render.DeviceScene(scene), torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda"))
    ds, out = ds_cache[name]
    st0 = torch.cuda.current_stream().cuda_stream
    ds.render_into(cam, p, out.data_ptr(), st0)
    ds.sync()
    best = 1e9
    for _ in range(reps):
        ds.render_into(cam, p, out.data_ptr(), st0)
        st = ds.sync()
        best = min(best, st.kernel_ms)
    return st.primary_rays / best / 1e3, st.node_tests / st.segments, float(out.double().sum())


for k in (1, 2):
    render.debug_set(capi.DEBUG_BVH_KERNEL, k)
    print(f"kernel {k}: " + "   ".join(f"{n} {bench(n)[0]:8.1f} Msamples/s" for n in scenes), flush=True)
sums = {n: bench(n)[2] for n in scenes}
render.debug_set(capi.DEBUG_BVH_KERNEL, 1)
assert all(bench(n, 1)[2] == sums[n] for n in scenes), "kernels disagree"
render.debug_set(capi.DEBUG_BVH_KERNEL, 2)
if "--sweep" in sys.argv:
    combos = [(sv, bl, sw, ks) for ks in (16, 28, 40) for sv in (32, 48) for bl in (4, 12) for sw in (2, 8)]
    combos += [(40, 10, 6, ks) for ks in (8, 20, 24, 32, 36, 44, 48, 56)]
    for sv, bl, sw, ks in combos:
        render.debug_set(capi.DEBUG_BVH2_KEEP, sv | (bl << 8) | (sw << 16) | (ks << 24))
        r = {n: bench(n, 2) for n in ("config3", "config5")}
        print(f"service {sv:2d} blocked {bl:2d} swap {sw:2d} stepping {ks:2d}: " +
              "   ".join(f"{n} {v[0]:8.1f} ({v[1]:.1f} boxes/seg)" for n, v in r.items()), flush=True)
