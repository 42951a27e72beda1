#!/usr/bin/env python3
"""Round-4 experiment: the walker / shader-wave BVH kernel (trace_kernel_bvhx, -DRAYZ_EXPERIMENTS builds, RAYZ_DEBUG_BVH_KERNEL = 3)
against the product kernel on configs 3, 5 and 2: bit-identical frames required, rates from the library's HIP events.
    bash tools/build_experiments.sh 4 && bash tools/with_lib.sh variants/lib_experiments_s4.so python tools/bvhx_bench.py [spp3 spp5 spp2] [--cfg=ns,xmin,batch,patience,prio ... (prio = shader | walker N << 2 | walker L/C << 4 | walker exchange << 6)]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer

render.init(0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfgs = [tuple(int(x) for x in a.split("=")[1].split(",")) for a in sys.argv[1:] if a.startswith("--cfg=")] or [(24, 12, 48, 8, 1)]
tops = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--top=")]
a = [int(x) for x in args[:3]] + [256, 128, 64][len(args[:3]):]


def frame(ds, cam, p, out, reps=3):
    st0 = torch.cuda.current_stream().cuda_stream
    ds.render_into(cam, p, out.data_ptr(), st0)
    ds.sync()
    best = 1e9
    for _ in range(reps):
        ds.render_into(cam, p, out.data_ptr(), st0)
        st = ds.sync()
        best = min(best, st.kernel_ms)
    return st, best


def bench(name, t, spp):
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    ref = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda")
    out = torch.empty_like(ref)
    ds = render.DeviceScene(scene)
    render.debug_set(capi.DEBUG_BVH_KERNEL, 1)
    st, ms = frame(ds, cam, p, ref)
    base = st.primary_rays / ms / 1e3
    print(f"{name}: product kernel {base:8.1f} Msamples/s  {ms:8.2f} ms  {st.node_tests / st.segments:.1f} box tests/seg  segs {st.segments}", flush=True)
    for cfg in cfgs:
        ns, xmin, batch, pat, prio = cfg
        render.debug_set(capi.DEBUG_BVH_KERNEL, 3)
        render.debug_set(capi.DEBUG_BVHX, ns | (xmin << 8) | (batch << 16) | (pat << 24) | (prio << 32))
        for top in (tops or [-1]):
            render.debug_set(capi.DEBUG_BVH_TOP, top)
            out.zero_()
            try:
                st2, ms2 = frame(ds, cam, p, out)
            except Exception as e:  # noqa: BLE001
                print(f"   exchange {cfg}: FAILED {e}", flush=True)
                continue
            same = bool(torch.equal(out, ref)) and st2.segments == st.segments
            r = st2.primary_rays / ms2 / 1e3
            print(f"   exchange slots {ns:2d} xmin {xmin:2d} batch {batch:2d} patience {pat:2d} prio {prio} top {top:5d}: {r:8.1f} Msamples/s ({100 * (r / base - 1):+5.1f} %)  {ms2:8.2f} ms  "
                  f"{st2.node_tests / st2.segments:.1f} box tests/seg  identical {same}", flush=True)
    render.debug_set(capi.DEBUG_BVH_KERNEL, -1)
    render.debug_set(capi.DEBUG_BVHX, -1)
    render.debug_set(capi.DEBUG_BVH_TOP, -1)
    ds.close()


bench("config3", tracer.randomBouncing(1920, -50, 50, seed=42), a[0])
if a[1]:
    bench("config5", tracer.triangleMesh(1920, 224, seed=1), a[1])
if a[2]:
    bench("config2", tracer.randomBouncing(1920, seed=42), a[2])
