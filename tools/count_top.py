import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from rayz_amd import capi, render, tracer
render.init(0)
for name, t, spp in (("config3", tracer.randomBouncing(1920, -50, 50, seed=42), 32), ("config5", tracer.triangleMesh(1920, 224, seed=1), 16), ("config2", tracer.randomBouncing(1920, seed=42), 32)):
    for top in (64, 256, 512, 1024, 1400):
        render.debug_set(capi.DEBUG_BVH_TOP, top)
        t.samples_per_px = spp
        t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
        scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
        out = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda")
        ds = render.DeviceScene(scene)
        print(name, "top", top, flush=True)
        sys.stderr.flush()
        ds.render_into(cam, p, out.data_ptr(), 0); ds.sync()
        ds.close()
