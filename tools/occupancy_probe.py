#!/usr/bin/env python3
"""How the one-path BVH kernel's rate depends on waves per SIMD: the same code, with unused LDS added to the workgroup's
request (rayz_hip_debug_set LDS_PAD) so that fewer 256-thread workgroups fit a CU (160 KB).  config 3, 256 spp."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer

render.init(0)
t = tracer.randomBouncing(1920, -50, 50, seed=42)
t.samples_per_px = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
ds = render.DeviceScene(scene)
out = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda")
st0 = torch.cuda.current_stream().cuda_stream
for kernel in (1, 2):
    render.debug_set(capi.DEBUG_BVH_KERNEL, kernel)
    for pad_kb, blocks in ((0, "max"), (10, 4), (22, 3), (48, 2), (100, 1)):
        render.debug_set(capi.DEBUG_LDS_PAD, pad_kb * 1024)
        best = 1e9
        for _ in range(3):
            ds.render_into(cam, p, out.data_ptr(), st0)
            st = ds.sync()
            best = min(best, st.kernel_ms)
        print(f"kernel {kernel}  LDS pad {pad_kb:3d} KB (<= {blocks} workgroups per CU): {st.primary_rays / best / 1e3:8.1f} Msamples/s", flush=True)
render.debug_set(capi.DEBUG_LDS_PAD, -1)
render.debug_set(capi.DEBUG_BVH_KERNEL, -1)
