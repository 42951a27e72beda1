#!/usr/bin/env python3
"""One wavefront render of config 3 (for rocprofv3 --kernel-trace --stats).  usage: wf_one.py [spp] [traversal 1|3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer
render.init(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
trav = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t = tracer.randomBouncing(1920, -50, 50, seed=42)
t.samples_per_px = spp
t.set_gpu(render_seed=1, traversal=trav)
scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
out = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda")
ds = render.DeviceScene(scene)
for _ in range(2):
    ds.render_into(cam, p, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = ds.sync()
print(st.primary_rays / st.kernel_ms / 1e3, "Msamples/s", st.kernel_ms, "ms", flush=True)
ds.close()
capi.load().rayz_hip_shutdown()
