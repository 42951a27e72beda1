import sys; sys.path.insert(0, "/root/repo")
from rayz_amd import capi, render, tracer
render.init(0)
t = tracer.randomBouncing(1920, -50, 50, seed=42)
t.samples_per_px = 32
t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
got, st = render.render_host(t.scene_desc(), t.camera_desc(), t.params())
print("STATS", st.primary_rays, st.segments, st.node_tests, st.sphere_tests, float(got.sum()))
