#!/usr/bin/env python3
"""BASELINE config 4 (10,003 spheres, 3840x2160, 4096 spp, dealt to 8 GPUs): every shard of the 8-way deal through the BVH, and three of
the eight through the flat list (17 s each), rendered on this one GPU.  Slowest shard = the frame time of the 8-GPU node before the
gather (99.5 MB in all; an estimated 0.2 ms over xGMI).   python tools/predicted_scaling_config4.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rayz_amd import capi, render, tracer
from rayz_amd import dist as rdist

render.init(0)
t = tracer.randomBouncing(3840, -50, 50, seed=42)
t.samples_per_px = 4096
ds = render.DeviceScene(t.scene_desc())
st0 = torch.cuda.current_stream().cuda_stream
total = 3840 * 2160 * 4096
for trav, name, ranks in ((capi.TRAVERSAL_BVH, "BVH", range(8)), (capi.TRAVERSAL_LINEAR, "flat list", (0, 3, 7))):
    t.set_gpu(render_seed=1, traversal=trav)
    cam, p0 = t.camera_desc(), t.params()
    ms, rate = [], []
    for rank in ranks:
        p = rdist.shard_params(p0, rank, 8)
        rows = render.shard_rows(p)
        out = torch.empty((rows, p.width, 3), dtype=torch.float32, device="cuda")
        if not ms:
            q = rdist.shard_params(p0, rank, 8)
            q.samples_per_px = 16
            ds.render_into(cam, q, out.data_ptr(), st0)  # uploads
            ds.sync()
        ds.render_into(cam, p, out.data_ptr(), st0)
        st = ds.sync()
        ms.append(st.kernel_ms)
        rate.append(st.primary_rays / st.kernel_ms / 1e3)
        print(f"  {name} shard {rank} of 8: {rows} rows, {st.kernel_ms:10.1f} ms, {rate[-1]:8.1f} Msamples/s on its GPU", flush=True)
    k = np.array(ms)
    print(f"config 4, {name}: slowest of the measured shards {k.max():.1f} ms -> {total / k.max() / 1e3:9.1f} Msamples/s on 8 GPUs "
          f"= {total / k.max() / 1e3 / np.mean(rate):.2f} x the mean per-GPU rate ({np.mean(rate):.1f})", flush=True)
ds.close()
