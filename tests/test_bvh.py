"""The reference's accelerator (src/hit.zig:44-217) as the product builds and traverses it.

CPU part: the product's host builder (rayz_amd/csrc/bvh_build.hpp, through rayz_hip_scene_bvh) against the
oracle's independent mode-A build, structural invariants, and the oracle's own BVH-vs-flat-list agreement.
GPU part (marked): the HIP traversal against the oracle's mode B with BVH traversal, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_images_equal
from rayz_amd import capi, tracer

_P = C.POINTER


def product_tree(t):
    lib = capi.load()
    sd = t.scene_desc()
    h = C.c_void_p()
    assert lib.rayz_hip_scene_create(C.byref(sd), C.byref(h)) == capi.OK
    n, depth = C.c_uint32(), C.c_uint32()
    assert lib.rayz_hip_scene_bvh(h, C.byref(n), C.byref(depth), None, None, None, None, None) == capi.OK
    N = n.value
    boxes = np.zeros((N, 6))
    skip, first, count = (np.zeros(N, np.uint32) for _ in range(3))
    order = np.zeros(sd.n_spheres + sd.n_triangles, np.uint32)
    assert lib.rayz_hip_scene_bvh(h, C.byref(n), None, boxes.ctypes.data_as(_P(C.c_double)),
                                  skip.ctypes.data_as(_P(C.c_uint32)), first.ctypes.data_as(_P(C.c_uint32)),
                                  count.ctypes.data_as(_P(C.c_uint32)), order.ctypes.data_as(_P(C.c_uint32))) == capi.OK
    lib.rayz_hip_scene_destroy(h)
    return dict(n=N, depth=depth.value, boxes=boxes, skip=skip, first=first, count=count, order=order)


def oracle_tree(oracle, t):
    lib = oracle.load()
    sd = t.scene_desc()
    N = lib.rayz_oracle_bvh_flat(C.byref(sd), None, None, None, None, None)
    boxes = np.zeros((N, 6))
    skip, first, count = (np.zeros(N, np.uint32) for _ in range(3))
    order = np.zeros(sd.n_spheres + sd.n_triangles, np.uint32)
    lib.rayz_oracle_bvh_flat(C.byref(sd), boxes.ctypes.data_as(_P(C.c_double)), skip.ctypes.data_as(_P(C.c_uint32)),
                             first.ctypes.data_as(_P(C.c_uint32)), count.ctypes.data_as(_P(C.c_uint32)),
                             order.ctypes.data_as(_P(C.c_uint32)))
    return dict(n=N, boxes=boxes, skip=skip, first=first, count=count, order=order)


def _scenes():
    yield "randomBouncing", tracer.randomBouncing(64, seed=42)
    yield "grid3", tracer.randomBouncing(64, -3, 3, seed=5)
    yield "10k", tracer.randomBouncing(64, -50, 50, seed=42)
    yield "three", tracer.threeSpheres(64, seed=1)
    yield "mesh", tracer.triangleMesh(64, 12, seed=1)
    t = tracer.Tracer.init(64, 20, 1, 0, (0, 0, 3), (0, 0, 0), (0, 1, 0), seed=1)
    tx = t.pool.add_solid_texture((0.5, 0.5, 0.5))
    m = t.pool.add_diffuse(tx)
    t.pool.add_sphere((0, 0, 0), 1.0, m)
    yield "one", t
    t = tracer.Tracer.init(64, 20, 1, 0, (0, 0, 3), (0, 0, 0), (0, 1, 0), seed=1)
    tx = t.pool.add_solid_texture((0.5, 0.5, 0.5))
    m = t.pool.add_diffuse(tx)
    for k in range(7):  # equal keys on every axis: exercises the STABLE sort (std.mem.sort)
        t.pool.add_sphere((0, 0, 0), 0.5 + 0.0 * k, m)
    yield "coincident", t


@pytest.mark.parametrize("name", ["randomBouncing", "grid3", "10k", "three", "one", "coincident", "mesh"])
def test_product_builder_equals_oracle_build(built, oracle, name):
    t = dict(_scenes())[name]
    a, b = product_tree(t), oracle_tree(oracle, t)
    assert a["n"] == b["n"]
    for k in ("boxes", "skip", "first", "count", "order"):
        assert np.array_equal(a[k], b[k]), k


def test_sphere_bbox_and_enclose_through_the_builder(built):
    """The reference's "sphere bbox" (src/geom.zig:69-84) and "enclose bbox" (src/hit.zig:237-245) vectors."""
    t = tracer.Tracer.init(64, 20, 1, 0, (0, 0, 3), (0, 0, 0), (0, 1, 0), seed=1)
    m = t.pool.add_dielectric(1.5)
    t.pool.add_sphere((0, 0, 0), 1.0, m)  # stationary unit sphere: [-1, 1]
    tr = product_tree(t)
    assert tr["n"] == 1 and tr["boxes"][0].tolist() == [-1, -1, -1, 1, 1, 1]
    t.pool.add_sphere((0, 0, 0), 1.0, m, velocity=(1, 1, 1))  # moving (0 -> 1): [-1, 2]
    tr = product_tree(t)
    assert tr["n"] == 1 and tr["count"][0] == 2
    assert tr["boxes"][0].tolist() == [-1, -1, -1, 2, 2, 2]  # union of the two boxes = enclose


def test_tree_invariants(built):
    t = tracer.randomBouncing(64, -20, 20, seed=3)
    tr = product_tree(t)
    N, ns = tr["n"], t.info().n_spheres
    leaves = tr["count"] > 0
    assert tr["count"].max() <= 2 and tr["count"][leaves].sum() == ns  # leaves of <= 2 own every hittable once
    assert N == 2 * leaves.sum() - 1
    assert sorted(tr["order"].tolist()) == list(range(ns))
    assert tr["skip"][0] == N and (tr["skip"] > np.arange(N)).all()
    # pre-order: an inner node's left child is the next node, its right child is the left child's skip
    for i in np.flatnonzero(~leaves)[:500]:
        l = i + 1
        r = tr["skip"][l]
        assert r < tr["skip"][i] and tr["skip"][r] == tr["skip"][i]
        for c in (l, r):  # children inside the parent box
            assert (tr["boxes"][c][:3] >= tr["boxes"][i][:3]).all() and (tr["boxes"][c][3:] <= tr["boxes"][i][3:]).all()
    assert 10 <= tr["depth"] <= 16  # median split: ceil(log2(n/2)) + 1


def test_oracle_bvh_traversal_equals_flat_list(oracle):
    """Mode B: the skip-link walk finds the same nearest hits as the flat list (same image, same segments) while
    testing ~37 boxes + ~5 spheres per segment instead of every sphere."""
    t = tracer.randomBouncing(96, seed=42)
    t.samples_per_px = 8
    t.set_gpu(render_seed=3, traversal=capi.TRAVERSAL_LINEAR)  # (485 hittables: AUTO would walk the BVH)
    sd, cam = t.scene_desc(), t.camera_desc()
    flat, sf = oracle.render_b(sd, cam, t.params())
    t.set_gpu(traversal=capi.TRAVERSAL_BVH)
    bvh, sb = oracle.render_b(sd, cam, t.params())
    assert np.array_equal(flat, bvh) and sf.segments == sb.segments
    assert sf.node_tests == 0
    # the oracle COUNTS its flat-list tests in the scan loop: every hittable, every segment (this identity pins the counter the
    # roofline flops in bench.py are priced from; the library derives the same figure from its segment counter)
    assert sf.sphere_tests == sf.segments * sd.n_spheres
    assert 25 < sb.node_tests / sb.segments < 50 and 3 < sb.sphere_tests / sb.segments < 8
    # and mode A's recursive traversal visits a comparable number of boxes (same tree, f64, first-hit order)
    pa = t.params()
    pa.precision, pa.tmin = capi.PRECISION_F64, 1e-10
    rs = t.rng_state().copy()
    _, sa = oracle.render_a(sd, cam, pa, rs)
    assert abs(sa.node_tests / sa.segments - sb.node_tests / sb.segments) < 2.0


# ---- GPU ----------------------------------------------------------------------------------------------
def _check_counters(gst, ost):
    """Segments are exact.  Box / primitive test counts are work done, not results: the GPU walks the same tree
    nearer-child-first with both child boxes tested per visit and parks leaves / candidates for a later phase, the
    oracle walks it left-then-right like the reference; the nearest hit is order-independent, the pruning is not."""
    assert gst.segments == ost.segments
    assert 0.3 * ost.node_tests <= gst.node_tests <= 2.0 * ost.node_tests + 64
    # (+ up to 8 oversized hittables kept out of the GPU's tree and tested once per segment, bvh_build.hpp)
    assert 0.3 * ost.sphere_tests <= gst.sphere_tests <= 2.0 * ost.sphere_tests + 8 * gst.segments + 64


def _pair(gpu, oracle, t):
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    got, gst = gpu.render_host(scene, cam, p)
    want, ost = oracle.render_b(scene, cam, p)
    return got, want, gst, ost


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [capi.PRECISION_F32, capi.PRECISION_F64])
def test_gpu_bvh_parity_random_bouncing(gpu, oracle, prec):
    t = tracer.randomBouncing(160, seed=42)
    t.samples_per_px = 16
    t.set_gpu(render_seed=5, traversal=capi.TRAVERSAL_BVH, precision=prec)
    got, want, gst, ost = _pair(gpu, oracle, t)
    assert_images_equal(got, want, f"BVH randomBouncing precision {prec}")
    _check_counters(gst, ost)


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["spheres", "mesh"])
def test_gpu_image_does_not_depend_on_the_walked_tree_or_its_record_format(gpu, oracle, scene):
    """The tree the GPU walks is the product's own (surface-area split, bvh_build.hpp) in one of two record formats (f32
    planes / 16-bit plane indices, DevScene::bvh_nodes): every combination — and the reference's median split — gives the
    oracle's image bit for bit, both precisions; the SAH tree needs fewer box tests."""
    t = tracer.randomBouncing(128, -30, 30, seed=42) if scene == "spheres" else tracer.triangleMesh(128, 40, seed=3)
    t.samples_per_px = 6
    tests = {}
    try:
        for prec in (capi.PRECISION_F32, capi.PRECISION_F64):
            t.set_gpu(render_seed=9, traversal=capi.TRAVERSAL_BVH, precision=prec)
            sd, cam, p = t.scene_desc(), t.camera_desc(), t.params()
            want, ost = oracle.render_b(sd, cam, p)
            for split in (0, 1):
                for fmt in (1, 2):
                    gpu.debug_set(capi.DEBUG_BVH_SPLIT, split)
                    gpu.debug_set(capi.DEBUG_BVH_NODES, fmt)
                    got, gst = gpu.render_host(sd, cam, p)  # (a fresh scene per call: the knobs act when the tree is built)
                    assert_images_equal(got, want, f"{scene} precision {prec} split {split} node format {fmt}")
                    assert gst.segments == ost.segments
                    tests[(prec, split, fmt)] = gst.node_tests
        for prec in (capi.PRECISION_F32, capi.PRECISION_F64):
            assert tests[(prec, 0, 1)] < 0.97 * tests[(prec, 1, 1)]  # SAH: fewer box tests than the median split
            assert abs(tests[(prec, 0, 2)] - tests[(prec, 0, 1)]) < 0.02 * tests[(prec, 0, 1)]  # index boxes: a cell larger at most
    finally:
        gpu.debug_set(capi.DEBUG_BVH_SPLIT, 0)
        gpu.debug_set(capi.DEBUG_BVH_NODES, 0)


@pytest.mark.gpu
def test_gpu_bvh_parity_10k_and_equals_flat_list(gpu, oracle):
    t = tracer.randomBouncing(128, -50, 50, seed=42)
    t.samples_per_px = 8
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
    got, want, gst, ost = _pair(gpu, oracle, t)
    assert_images_equal(got, want, "BVH 10k spheres")
    _check_counters(gst, ost)
    t.set_gpu(traversal=capi.TRAVERSAL_LINEAR)
    flat, fst = gpu.render_host(t.scene_desc(), t.camera_desc(), t.params())
    # the box test is conservative and the nearest hit (with its tie rule) order-independent: the SAME image.  (Until
    # the slab test's reciprocal was capped this held only up to "a handful of pixels", which hid the zero-component
    # bug of DESIGN.md 4.8.)
    assert np.array_equal(flat, got) and fst.segments == gst.segments


@pytest.mark.gpu
def test_lds_request_falls_back_to_a_shorter_top(gpu, oracle):
    """render_impl sizes the tree's LDS top for a 150 KB workgroup; a stack that refuses the request must not fail every BVH
    render: the launch retries with a shorter prefix of the top (the walk reads the rest from global memory).  Forced here with
    RAYZ_DEBUG_LDS_PAD (unused bytes added to the request): same image as the oracle, a pad no retry can absorb is an error."""
    from rayz_amd import render

    t = tracer.randomBouncing(96, -20, 20, seed=42)
    t.samples_per_px, t.max_bounces = 4, 10
    t.set_gpu(render_seed=9, traversal=capi.TRAVERSAL_BVH)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    want, ost = oracle.render_b(scene, cam, p)
    try:
        for pad in (0, 8 * 1024, 40 * 1024, 70 * 1024):
            render.debug_set(capi.DEBUG_LDS_PAD, pad)
            got, gst = gpu.render_host(scene, cam, p)
            assert_images_equal(got, want, f"LDS pad {pad}")
            assert gst.segments == ost.segments
        render.debug_set(capi.DEBUG_LDS_PAD, 159 * 1024)  # more than a CU has beside the stacks, whatever the top
        with pytest.raises(capi.RayzHipError, match="LDS"):
            gpu.render_host(scene, cam, p)
    finally:
        render.debug_set(capi.DEBUG_LDS_PAD, -1)


@pytest.mark.gpu
def test_gpu_bvh_equals_flat_list_on_the_whole_config3_frame(gpu):
    """1920x1080, 10,003 spheres, 16 spp = 33 M paths, ≈1e8 segments through both traversals on the device: identical
    images, identical segment counts."""
    t = tracer.randomBouncing(1920, -50, 50, seed=42)
    t.samples_per_px = 16
    t.set_gpu(render_seed=5, traversal=capi.TRAVERSAL_BVH)
    bvh, bst = gpu.render_host(t.scene_desc(), t.camera_desc(), t.params())
    t.set_gpu(traversal=capi.TRAVERSAL_LINEAR)
    flat, fst = gpu.render_host(t.scene_desc(), t.camera_desc(), t.params())
    assert np.array_equal(bvh, flat) and bst.segments == fst.segments and np.isfinite(bvh).all()


@pytest.mark.gpu
def test_gpu_bvh_equals_flat_list_at_baseline_config3_in_full(gpu, oracle):
    """BASELINE.json configs[2] exactly — 10,003 spheres, 1920x1080, 1024 spp, 50 bounces: 2.1e9 paths, 6.3e9 segments
    through the flat list (the headline kernel, ~9 s) and through the BVH on the device: the same image bit for bit, the
    same segment count, no non-finite pixel."""
    t = tracer.randomBouncing(1920, -50, 50, seed=42)
    t.samples_per_px = 1024
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
    bvh, bst = gpu.render_host(t.scene_desc(), t.camera_desc(), t.params())
    t.set_gpu(traversal=capi.TRAVERSAL_LINEAR)
    flat, fst = gpu.render_host(t.scene_desc(), t.camera_desc(), t.params())
    assert np.array_equal(bvh, flat) and bst.segments == fst.segments and np.isfinite(flat).all()
    assert fst.primary_rays == 1920 * 1080 * 1024 and 2.9 < fst.segments / fst.primary_rays < 3.0
    # .. and 32 scattered pixels of it by the oracle at the full 1024 spp (through its own BVH walk: the same nearest hits), bit for
    # bit: pins the summation tree of the metric's own schedule (64 x 15, 32, 16, 16: DESIGN.md 4.6) on the metric's own frame
    from helpers import assert_images_equal

    rng = np.random.default_rng(4)
    pix = np.unique(np.concatenate([rng.integers(0, 1920 * 1080, 28), [0, 1919, 1920 * 1079, 1920 * 1080 - 1]])).astype(np.uint32)
    t.set_gpu(traversal=capi.TRAVERSAL_BVH)
    want, _ = oracle.render_b(t.scene_desc(), t.camera_desc(), t.params(), pixels=pix)
    assert_images_equal(flat.reshape(-1, 3)[pix], want, "config 3 at full size and full spp, oracle spot pixels")


@pytest.mark.gpu
def test_baseline_config2_in_full(gpu, oracle):
    """BASELINE.json configs[1] exactly as bench.py measures it — `randomBouncing` as the reference ships it (src/rayz.zig:45-168:
    485 spheres, scene seed 42), 1920x1080, 256 spp, 50 bounces, render seed 1: the flat list and the BVH on the device give the
    same image bit for bit and the same segment count; 48 scattered pixels rendered by the oracle at the full 256 spp match bit for
    bit; the frame is a usable image.  (The only BASELINE config the suite used to run in miniature only.)"""
    t = tracer.randomBouncing(1920, seed=42)
    assert (t.info().width, t.info().height, t.info().n_spheres) == (1920, 1080, 485)
    t.samples_per_px, t.max_bounces = 256, 50
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    bvh, bst = gpu.render_host(scene, cam, p)
    t.set_gpu(traversal=capi.TRAVERSAL_LINEAR)
    pf = t.params()
    flat, fst = gpu.render_host(scene, cam, pf)
    assert np.array_equal(bvh, flat) and bst.segments == fst.segments
    assert flat.shape == (1080, 1920, 3) and np.isfinite(flat).all() and (flat >= 0).all()
    assert fst.primary_rays == 1920 * 1080 * 256 and 2.6 < fst.segments / fst.primary_rays < 2.9
    rng = np.random.default_rng(2)
    pix = np.unique(np.concatenate([rng.integers(0, 1920 * 1080, 44), [0, 1919, 1920 * 1079, 1920 * 1080 - 1]])).astype(np.uint32)
    want, _ = oracle.render_b(scene, cam, pf, pixels=pix)
    assert_images_equal(flat.reshape(-1, 3)[pix], want, "config 2 at full size, oracle spot pixels")
    t.set_gpu(traversal=capi.TRAVERSAL_AUTO)  # what Tracer.render() uses: the BVH for this scene (485 > 160 hittables)
    auto, ast = gpu.render_host(scene, cam, t.params())
    assert np.array_equal(auto, flat) and ast.node_tests > 0


@pytest.mark.gpu
def test_gpu_f64_mode_bvh_equals_flat_list_on_whole_frames(gpu):
    """The f64 fidelity mode (f32 filters, f64 decisions): config 3 at 64 spp and the 100k-triangle mesh of config 5 at
    4 spp, flat list against BVH on the device — identical images and segment counts."""
    for t, spp in ((tracer.randomBouncing(1920, -50, 50, seed=42), 64), (tracer.triangleMesh(1920, 224, seed=1), 4)):
        t.samples_per_px = spp
        t.set_gpu(render_seed=2, traversal=capi.TRAVERSAL_BVH, precision=capi.PRECISION_F64)
        bvh, bst = gpu.render_host(t.scene_desc(), t.camera_desc(), t.params())
        t.set_gpu(traversal=capi.TRAVERSAL_LINEAR)
        flat, fst = gpu.render_host(t.scene_desc(), t.camera_desc(), t.params())
        assert bvh.dtype == np.float64 and np.array_equal(bvh, flat) and bst.segments == fst.segments and np.isfinite(flat).all()


@pytest.mark.gpu
def test_gpu_bvh_edge_cases(gpu, oracle):
    for name in ("one", "coincident", "three"):
        t = dict(_scenes())[name]
        t.samples_per_px, t.max_bounces = 4, 6
        t.set_gpu(render_seed=2, traversal=capi.TRAVERSAL_BVH)
        got, want, gst, ost = _pair(gpu, oracle, t)
        assert_images_equal(got, want, f"BVH {name}")
        _check_counters(gst, ost)
    # empty pool: no tree at all, background only
    t = tracer.Tracer.init(48, 40.0, 1.0, 0.0, (0, 0, 0), (0, 0.3, -1), (0, 1, 0), seed=1)
    t.samples_per_px = 2
    t.set_gpu(render_seed=2, traversal=capi.TRAVERSAL_BVH)
    got, want, gst, _ = _pair(gpu, oracle, t)
    assert_images_equal(got, want, "BVH empty scene")
    assert gst.node_tests == 0 and gst.segments == gst.primary_rays


@pytest.mark.gpu
def test_gpu_bvh_shards_and_custom_scene(gpu, oracle):
    from test_gpu_parity import _custom_scene

    t = _custom_scene()
    t.samples_per_px, t.max_bounces = 12, 20
    t.set_gpu(render_seed=9, traversal=capi.TRAVERSAL_BVH)
    got, want, gst, ost = _pair(gpu, oracle, t)
    assert_images_equal(got, want, "BVH custom scene")
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    out = np.zeros_like(got)
    for idx in range(3):
        q = t.params()
        q.tile_rows, q.shard_index, q.shard_count = 4, idx, 3
        part, _ = gpu.render_host(scene, cam, q)
        out[gpu.shard_row_indices(p.height, 4, idx, 3)] = part
    assert_images_equal(out, got, "BVH 3 shards")


@pytest.mark.gpu
def test_gpu_config4_full_frame_on_one_gpu(gpu, oracle):
    """BASELINE configs[3] (3840x2160, 4096 spp, 10,003 spheres) in full on ONE GPU through the BVH kernel: 2.1e9
    work items behind a 32-bit queue and a 2.65 GB chunk-sum workspace (20 chunk sums per pixel with the schedule of DESIGN.md §4.6).  16 scattered
    pixels are checked bit for bit against the oracle at the same 4096 spp; counters obey their identities."""
    t = tracer.randomBouncing(3840, -50, 50, seed=42)
    t.samples_per_px = 4096
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_BVH)
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    assert (p.width, p.height) == (3840, 2160)
    got, st = gpu.render_host(scene, cam, p)
    assert got.shape == (2160, 3840, 3) and np.isfinite(got).all() and (got >= 0).all()
    assert st.primary_rays == 3840 * 2160 * 4096
    assert 2.0 < st.segments / st.primary_rays < 4.5 and st.node_tests > 20 * st.segments
    rng = np.random.default_rng(4)
    pix = np.unique(np.concatenate([rng.integers(0, 3840 * 2160, 14), [0, 3840 * 2160 - 1]])).astype(np.uint32)
    want, _ = oracle.render_b(scene, cam, p, pixels=pix)
    assert_images_equal(got.reshape(-1, 3)[pix], want, "config 4 spot pixels")
    assert got[:80].mean() > got[-80:].mean()
