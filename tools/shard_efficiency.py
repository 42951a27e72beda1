#!/usr/bin/env python3
"""How well does ONE GPU's share of an 8-way frame use the GPU?  (round 4)  Config 3 through the BVH and through the flat list:
whole frame vs shard 0 of 8 (8-row tiles), at the automatic chunk schedule and at uniform chunks (chunk_spp) — a shard has ~1 pixel per
lane, so the size of the work items decides how evenly the launch ends.  (Negative chunk values = the automatic schedule capped at that
chunk: RAYZ_DEBUG_CHUNK_CAP, -DRAYZ_EXPERIMENTS builds only — bash tools/build_experiments.sh; the product build skips them.)
    python tools/shard_efficiency.py [spp_bvh spp_flat]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayz_amd import capi, render, tracer
from rayz_amd import dist as rdist

render.init(0)
a = [int(x) for x in sys.argv[1:3]] + [1024, 64][len(sys.argv[1:3]):]
t = tracer.randomBouncing(1920, -50, 50, seed=42)
ds = render.DeviceScene(t.scene_desc())
st0 = torch.cuda.current_stream().cuda_stream
for trav, spp, name in ((capi.TRAVERSAL_BVH, a[0], "bvh"), (capi.TRAVERSAL_LINEAR, a[1], "flat")):
    t.samples_per_px = spp
    t.set_gpu(render_seed=1, traversal=trav)
    cam, p0 = t.camera_desc(), t.params()
    for chunk in ((0, -64, -32) if (trav == capi.TRAVERSAL_LINEAR and spp >= 512) else (0, -128, -64, -32, 32, 64)):  # < 0: the automatic (graded) schedule capped at that chunk
        res = []
        for world in (1, 8):
            p = rdist.shard_params(p0, 0, world)
            p.chunk_spp = max(chunk, 0)
            try:
                render.debug_set(capi.DEBUG_CHUNK_CAP, -chunk if chunk < 0 else -1)
            except capi.RayzHipError:
                if chunk < 0:
                    res = None
                    break
            rows = render.shard_rows(p)
            out = torch.empty((rows, p.width, 3), dtype=torch.float32, device="cuda")
            best = 1e9
            for _ in range(1 if (trav == capi.TRAVERSAL_LINEAR and spp >= 512) else 3):
                ds.render_into(cam, p, out.data_ptr(), st0)
                st = ds.sync()
                best = min(best, st.kernel_ms)
            res.append((best, rows * p.width * spp / best / 1e3))
        if not res:
            continue
        print(f"{name} {spp} spp chunk_spp {chunk:3d}: whole frame {res[0][0]:9.2f} ms {res[0][1]:8.1f} Msamples/s | shard 0 of 8 {res[1][0]:9.2f} ms {res[1][1]:8.1f} Msamples/s "
              f"= {100 * res[1][1] / res[0][1]:5.1f} % of the whole-frame rate", flush=True)
ds.close()
