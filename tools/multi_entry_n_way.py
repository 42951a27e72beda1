#!/usr/bin/env python3
"""The C ABI's one-call multi-device entry (rayz_hip_multi_render) with N "devices" that are all device 0 (the test-only duplicate flag,
peer copies) on BASELINE config 3 through the BVH at full size: N scenes, the N-way deal, the gather into N slots and the un-interleave
run for real; the frame must hash like the single-device frame.  Per-device kernel times are each shard's own (they run one after
the other or overlapped on the one GPU — no scaling figure).   python tools/multi_entry_n_way.py [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rayz_amd import capi, render, tracer
from rayz_amd import dist as rdist

render.init(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
t = tracer.randomBouncing(1920, -50, 50, seed=42)
t.samples_per_px = spp
t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
one, st1 = render.render_host(scene, cam, p)
h1 = rdist.frame_sha256(torch.from_numpy(one))
print(f"single device: {st1.kernel_ms:.1f} ms, frame_sha256 {h1[:16]}", flush=True)
for n in (2, 4, 8):
    m = render.MultiScene(scene, [0] * n, capi.GATHER_PEER_COPY | capi.GATHER_ALLOW_DUPLICATE_DEVICES)
    m.render(cam, p)
    t0 = time.perf_counter()
    frame, st = m.render(cam, p)
    dt = time.perf_counter() - t0
    per = [d.kernel_ms for d in m.device_stats()]
    g, f = m.timing()
    h = rdist.frame_sha256(torch.from_numpy(frame))
    print(f"N = {n} (all device 0): frame_sha256 {h[:16]} {'== single device' if h == h1 else 'DIFFERS'}; host-to-host {dt * 1e3:.1f} ms; per-device kernel ms "
          f"{' '.join(f'{x:.1f}' for x in per)}; gather {g:.2f} ms, gather + copy-out {f:.2f} ms; segments {st.segments} ({'==' if st.segments == st1.segments else '!='})", flush=True)
    m.close()
