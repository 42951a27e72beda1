"""Mode B — what every `-m gpu` test compares the kernels with — pinned directly:

* against the reference's own test vectors (src/material.zig:213-223 refract, src/renderer.zig:129-149 get ray,
  src/hit.zig:247-279 bbox hit), in f32 and f64, piece by piece (rayz_oracle_kat_b evaluates the functions
  tracePath is assembled from);
* against mode A (the line-by-line restatement of the reference) DETERMINISTICALLY: the same (ray, sphere) pairs,
  camera rays, scatter events — with SHARED uniforms — and boxes through the reference's functions and through
  mode B's; same decisions, values within the stated bounds.  10^6 sphere pairs and 10^6 scatter events.

tests/test_kat_gpu.py holds the HIP device functions to the same records bit for bit."""
import numpy as np
import pytest

import kat_records as K
from rayz_amd import capi

F32, F64 = capi.PRECISION_F32, capi.PRECISION_F64


# ---- reference vectors ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec,rel", [(F32, 1e-4), (F64, 1e-4)])
def test_mode_b_refract_reference_vector(oracle, prec, rel):
    rec, want = K.refract_reference()
    got = oracle.kat_b(capi.KAT_REFRACT, rec, prec)[0, :3]
    assert got == pytest.approx(want, rel=rel)  # the reference's own tolerance (expectApproxEqRel 0.0001)
    assert got == pytest.approx(oracle.kat_a(capi.KAT_REFRACT, rec)[0, :3], abs=3e-7 if prec == F32 else 1e-15)


@pytest.mark.parametrize("prec,rel", [(F32, 1e-5), (F64, 1e-5)])
def test_mode_b_get_ray_reference_vector(oracle, prec, rel):
    cam = capi.CameraDesc()
    oracle.load().rayz_oracle_camera_init(90, 12 ** 0.5, 0, oracle.d3([-2, 2, 1]), oracle.d3([0, 0, -1]),
                                          oracle.d3([0, 1, 0]), 225, 400, cam)
    rec, want = K.get_ray_reference(cam)
    got = oracle.kat_b(capi.KAT_GET_RAY, rec, prec)
    assert got[:, 0:3].tolist() == [[-2, 2, 1]] * 2  # origin = look_from
    assert got[:, 3:6] == pytest.approx(want, rel=rel)
    assert got[:, 7].tolist() == [3, 3]  # jitter x, jitter y, time (no lens draws: defocus 0)


@pytest.mark.parametrize("prec", [F32, F64])
def test_mode_b_box_hit_reference_vectors(oracle, prec):
    rec, want = K.box_hit_reference()
    assert oracle.kat_b(capi.KAT_BOX_HIT, rec, prec)[:, 0].tolist() == want.tolist()
    assert oracle.kat_a(capi.KAT_BOX_HIT, rec)[:, 0].tolist() == want.tolist()


def test_product_sphere_bbox_reference_vector(built):
    """src/geom.zig:69-84 through the PRODUCT's host builder (rayz_amd/csrc/bvh_build.hpp): the root box of a
    one-sphere pool is that sphere's bbox."""
    import ctypes as C

    lib = capi.load()
    tex = (capi.Texture * 1)(capi.Texture(kind=capi.TEX_SOLID, color=capi.D3(1, 1, 1)))
    mat = (capi.Material * 1)(capi.Material(kind=capi.MAT_DIFFUSE, texture=0, method=2))
    for vel, lo, hi in [((0, 0, 0), [-1] * 3, [1] * 3), ((1, 1, 1), [-1] * 3, [2] * 3)]:
        sph = (capi.Sphere * 1)(capi.Sphere(center=capi.D3(0, 0, 0), velocity=capi.D3(*vel), radius=1.0, material=0))
        sd = capi.SceneDesc(spheres=sph, materials=mat, textures=tex, n_spheres=1, n_materials=1, n_textures=1)
        h = C.c_void_p()
        assert lib.rayz_hip_scene_create(C.byref(sd), C.byref(h)) == capi.OK
        n = C.c_uint32()
        boxes = (C.c_double * 6)()
        assert lib.rayz_hip_scene_bvh(h, C.byref(n), None, boxes, None, None, None, None) == capi.OK
        assert n.value == 1 and list(boxes) == lo + hi
        lib.rayz_hip_scene_destroy(h)


# ---- mode A vs mode B, deterministic ------------------------------------------------------------------------------
def _sphere_margins(rec):
    """Distance of each pair from the decisions' borders, from the inputs in f64: relative discriminant and the
    margins of the roots against tmin."""
    c, v, r, o, d, time = rec[:, 0:3], rec[:, 3:6], rec[:, 6], rec[:, 7:10], rec[:, 10:13], rec[:, 13]
    q = c + v * time[:, None] - o
    a, hb = (d * d).sum(1), (d * q).sum(1)
    cc = (q * q).sum(1) - r * r
    disc = hb * hb - a * cc
    scale = hb * hb + a * np.abs((q * q).sum(1)) + a * r * r
    rt = np.sqrt(np.maximum(disc, 0))
    t1, t2 = (hb - rt) / a, (hb + rt) / a
    # size of the terms a root is the difference of, + what the cancellation in c + v t - o costs ((|c| + |o|) / |d|)
    tscale = (np.abs(hb) + rt) / a + (np.linalg.norm(c, axis=1) + np.linalg.norm(o, axis=1)) / np.sqrt(a)
    return disc / scale, np.minimum(np.abs(t1 - rec[:, 14]), np.abs(t2 - rec[:, 14])) / (1 + np.abs(t1) + np.abs(t2)), tscale


@pytest.mark.parametrize("big", [False, True])
def test_sphere_hit_mode_b_equals_mode_a_on_a_million_pairs(oracle, big):
    rng = np.random.default_rng(11 + big)
    n_total = 1_000_000 if not big else 200_000
    worst = {"t64": 0.0, "t32": 0.0, "n32": 0.0, "p32": 0.0}
    hits = borderline = 0
    for _ in range(n_total // 100_000):
        rec = K.random_sphere_hits(rng, 100_000, big)
        a = oracle.kat_a(capi.KAT_SPHERE_HIT, rec)
        b64 = oracle.kat_b(capi.KAT_SPHERE_HIT, rec, F64)
        b32 = oracle.kat_b(capi.KAT_SPHERE_HIT, rec, F32)
        drel, tmar, tscale = _sphere_margins(rec)
        # the f64 evaluation differs from the reference's only by FMA contraction: decisions equal unless the
        # discriminant (relative to its terms) or a root's distance from tmin is within rounding
        clear64 = (np.abs(drel) > 1e-13) & (tmar > 1e-13)
        assert (a[clear64, 0] == b64[clear64, 0]).all()
        # conservative filter: a pair the reference hits is never filtered out, in either precision
        assert (b64[a[:, 0] == 1, 9] == 1).all() and (b32[a[:, 0] == 1, 9] == 1).all()
        both = clear64 & (a[:, 0] == 1)
        # a root is (hb -+ sqrt(disc)) / a: rounding of hb and of the quotient, plus the discriminant's rounding
        # (relative to its terms) amplified by 1 / (2 drel) through the square root
        et = np.abs(a[both, 1] - b64[both, 1]) / tscale[both]
        assert (et <= 2.3e-16 * (4 + 1.0 / np.abs(drel[both]))).all(), et.max()  # 2.3e-16 = 2 ulp steps
        assert np.abs(a[both, 2:8] - b64[both, 2:8]).max() <= 1e-9 * (1 + np.abs(rec[both, 0:3]).max())
        assert (a[both, 8] == b64[both, 8]).all()
        worst["t64"] = max(worst["t64"], float(et.max()))
        # f32 path state, f64 roots rounded to f32: same decision away from the borders, t within f32 rounding
        clear32 = (np.abs(drel) > 1e-6) & (tmar > 1e-6)
        assert (a[clear32, 0] == b32[clear32, 0]).all()
        both = clear32 & (a[:, 0] == 1)
        et = np.abs(a[both, 1] - b32[both, 1]) / np.maximum(np.abs(a[both, 1]), 1e-3)
        assert et.max() <= 1.2e-7, et.max()  # (R) rounding of the root: half an ulp = 6e-8, + the f64 differences
        scale = 1 + np.abs(rec[both, 7:10]).max(1) + np.abs(a[both, 1]) * np.abs(rec[both, 10:13]).max(1)
        ep = np.abs(a[both, 2:5] - b32[both, 2:5]).max(1) / scale
        assert ep.max() <= 4e-7, ep.max()
        # normal = (p - c) / r: the rounding of p and c (relative to r) plus the normalisation's own
        en = np.abs(a[both, 5:8] - b32[both, 5:8]).max(1) / (1 + (scale + np.abs(rec[both, 0:3]).max(1)) / rec[both, 6])
        assert en.max() <= 4e-7, en.max()
        worst["t32"], worst["p32"], worst["n32"] = max(worst["t32"], float(et.max())), max(worst["p32"], float(ep.max())), max(worst["n32"], float(en.max()))
        hits += int(a[:, 0].sum())
        borderline += int((~clear32).sum())
    assert 0.25 * n_total < hits < 0.85 * n_total and borderline < 0.01 * n_total
    print(f"sphere pairs {n_total}: {hits} hits, {borderline} borderline (excluded for f32), worst {worst}")


def test_scatter_mode_b_equals_mode_a_on_a_million_events_with_shared_uniforms(oracle):
    rng = np.random.default_rng(5)
    n_total, mism32 = 1_000_000, 0
    for _ in range(n_total // 100_000):
        rec = K.random_scatters(rng, 100_000)
        a = oracle.kat_a(capi.KAT_SCATTER, rec)
        rec = rec[a[:, 4] <= rec[:, 16]]  # (a rejection loop that outran the record's list of uniforms: ~1e-4 of the events)
        a = oracle.kat_a(capi.KAT_SCATTER, rec)
        assert len(rec) > 99_900
        for prec, tol in ((F64, 2e-12), (F32, 3e-5)):
            b = oracle.kat_b(capi.KAT_SCATTER, rec, prec)
            same = (a[:, 0] == b[:, 0]) & (a[:, 4] == b[:, 4])  # scattered / absorbed, and number of draws consumed
            if prec == F64:
                # f64: the only differences are FMA contraction and x^5 by multiplies; a decision can flip only within
                # ~1e-15 of its border, which 10^6 random events do not reach
                assert same.all(), np.flatnonzero(~same)[:5]
            else:
                mism32 += int((~same).sum())  # rejection / Schlick / absorption decisions within f32 rounding of their border
            ok = same & (a[:, 0] == 1)
            scale = 1 + np.abs(rec[ok, 6:9]).max(1) + np.abs(rec[ok, 9:12]).max(1)  # |d| and the diffuse target POINT
            err = np.abs(a[ok, 1:4] - b[ok, 1:4]).max(1) / scale
            assert err.max() <= tol, (prec, err.max(), np.flatnonzero(ok)[err.argmax()])
    assert mism32 <= 20, mism32  # expected ~1e-7 per decision: a handful in 10^6 events x a few decisions each


def test_camera_ray_mode_b_equals_mode_a_with_shared_uniforms(oracle):
    rng = np.random.default_rng(2)
    cam = capi.CameraDesc()
    oracle.load().rayz_oracle_camera_init(20, 10.0, 0.6, oracle.d3([13, 2, 3]), oracle.d3([0, 0, 0]),
                                          oracle.d3([0, 1, 0]), 1080, 1920, cam)  # randomBouncing's camera
    rec = K.random_get_rays(rng, 200_000, cam)
    a = oracle.kat_a(capi.KAT_GET_RAY, rec)
    b64, b32 = oracle.kat_b(capi.KAT_GET_RAY, rec, F64), oracle.kat_b(capi.KAT_GET_RAY, rec, F32)
    assert (a[:, 7] == b64[:, 7]).all() and (a[:, 6] == b64[:, 6]).all()  # same number of lens tries, same time
    assert np.abs(a[:, 0:6] - b64[:, 0:6]).max() <= 1e-13 * 20
    same = a[:, 7] == b32[:, 7]
    assert (~same).sum() <= 2  # a lens sample within f32 rounding of the unit circle
    assert np.abs(a[same, 0:6] - b32[same, 0:6]).max() <= 2e-6 * 13


def test_refract_reflectance_checker_background_box_mode_b_equals_mode_a(oracle):
    rng = np.random.default_rng(3)
    rec = K.random_refracts(rng, 200_000)
    a = oracle.kat_a(capi.KAT_REFRACT, rec)
    assert np.isfinite(a).all()
    assert np.abs(a - oracle.kat_b(capi.KAT_REFRACT, rec, F64)).max() <= 2e-14
    assert np.abs(a - oracle.kat_b(capi.KAT_REFRACT, rec, F32)).max() <= 3e-6
    r2 = K.blank(200_000)
    r2[:, 0], r2[:, 1] = K.f32r(rng.uniform(0, 1, 200_000)), K.f32r(rng.uniform(0.4, 2.5, 200_000))
    a = oracle.kat_a(capi.KAT_REFLECTANCE, r2)[:, 0]  # std.math.pow(x, 5) vs x²·x²·x: DESIGN.md 4.5 says <= 2 ulp
    assert (np.abs(a - oracle.kat_b(capi.KAT_REFLECTANCE, r2, F64)[:, 0]) <= 4 * np.spacing(a)).all()
    assert np.abs(a - oracle.kat_b(capi.KAT_REFLECTANCE, r2, F32)[:, 0]).max() <= 3e-7
    r3 = K.random_checkers(rng, 200_000)
    a = oracle.kat_a(capi.KAT_CHECKER, r3)[:, 0]
    assert (a == oracle.kat_b(capi.KAT_CHECKER, r3, F64)[:, 0]).all()
    q = r3[:, 0:3] / r3[:, 3:4]
    clear = (np.abs(q - np.round(q)) > 1e-4).all(1)  # away from cell borders the f32 quotient floors the same way
    assert (a[clear] == oracle.kat_b(capi.KAT_CHECKER, r3, F32)[clear, 0]).all() and clear.mean() > 0.99
    r4 = K.blank(100_000)
    r4[:, 0:3] = K.f32r(rng.normal(size=(100_000, 3)) * rng.uniform(0.1, 10, (100_000, 1)))
    a = oracle.kat_a(capi.KAT_BACKGROUND, r4)[:, :3]
    assert np.abs(a - oracle.kat_b(capi.KAT_BACKGROUND, r4, F64)[:, :3]).max() <= 1e-15 * 4
    assert np.abs(a - oracle.kat_b(capi.KAT_BACKGROUND, r4, F32)[:, :3]).max() <= 4e-7
    # boxes: mode B's slab test is CONSERVATIVE (4-ulp slack, inclusive): it never misses a box the reference's
    # strict test hits; the other way round only at the border
    r5 = K.random_boxes(rng, 300_000)
    a = oracle.kat_a(capi.KAT_BOX_HIT, r5)[:, 0]
    for prec in (F64, F32):
        b = oracle.kat_b(capi.KAT_BOX_HIT, r5, prec)[:, 0]
        assert (b[a == 1] == 1).all()
        assert ((b == 1) & (a == 0)).mean() < 1e-3
    assert 0.2 < a.mean() < 0.8


def test_box_test_with_zero_direction_components(oracle):
    """A direction component of exactly 0 (a diffuse scatter produces one about once in 1e5 bounces at |p| = 50): with
    1/d = ±inf the one-FMA slab distance of a box that straddles 0 is −inf for one plane and NaN for the other, and the
    box was culled although the ray lies inside its slab (found by tools/fuzz_more.py as flat list != BVH).  The
    reciprocal is held to ±2^64 / ±2^512: mode B never misses a box the geometry hits, nor one mode A's literal
    reference test hits."""
    rng = np.random.default_rng(12)
    rec, truth, beside = K.axis_parallel_boxes(rng, 200_000)
    a = oracle.kat_a(capi.KAT_BOX_HIT, rec)[:, 0]
    assert (a[truth] == 1).all() and 0.1 < truth.mean() < 0.9 and beside.mean() > 0.2
    for prec in (F64, F32):
        b = oracle.kat_b(capi.KAT_BOX_HIT, rec, prec)[:, 0]
        assert (b[truth] == 1).all() and (b[a == 1] == 1).all()
        # still a test on the parallel axis: a ray that runs beside the slab is culled.  (On the OTHER axes such a ray
        # is no longer culled: the absolute slack carries 4u·K·|o_k| — the price of staying conservative for one ray in
        # 1e5; it then visits every box whose parallel slabs contain it.)
        assert (b[beside] == 0).all()
    # the hand case: box [-1, 2]^3 around the origin, ray inside it along +z with d.x = d.y = 0
    one = K.blank(4)
    one[:, 0:3], one[:, 3:6] = -1.0, 2.0
    one[:, 6:9] = [[0.5, 0.5, -5.0], [-0.5, 0.5, -5.0], [0.5, 2.5, -5.0], [-3.0, 0.5, -5.0]]
    one[:, 9:12] = [[0.0, 0.0, 1.0], [-0.0, 0.0, 2.0], [0.0, 0.0, 1.0], [0.0, -0.0, 1.0]]
    one[:, 12], one[:, 13] = 1e-3, np.inf
    for prec in (F64, F32):
        assert oracle.kat_b(capi.KAT_BOX_HIT, one, prec)[:, 0].tolist() == [1, 1, 0, 0]


def test_flat_list_equals_bvh_on_the_fuzz_scenes_that_found_the_zero_component_bug(oracle):
    """tests/test_fuzz_gpu.random_scene seeds on which mode B's BVH walk lost hits of the ground sphere (rays inside a
    dielectric r = 500 ground with an exactly-zero direction component): flat list and BVH agree bit for bit."""
    from test_fuzz_gpu import axis_scene, random_scene
    scenes = [random_scene(s) for s in (5003, 5008, 5010)] + [random_scene(s) for s in range(7000, 7030)] + \
             [axis_scene(s) for s in range(900, 920)]
    for seed, t in enumerate(scenes):
        got = {}
        for trav in (capi.TRAVERSAL_LINEAR, capi.TRAVERSAL_BVH):
            t.set_gpu(traversal=trav, precision=F32)
            got[trav] = oracle.render_b(t.scene_desc(), t.camera_desc(), t.params())
        a, b = got[capi.TRAVERSAL_LINEAR], got[capi.TRAVERSAL_BVH]
        assert np.array_equal(a[0], b[0], equal_nan=True) and a[1].segments == b[1].segments, seed


def test_triangle_mode_b_vs_mode_a(oracle):
    """Build-defined primitive (parity unpinned): f64 mode B decides like the literal f64 Möller–Trumbore except at
    edges within rounding; the f32 decision differs there by construction (the filter IS the test)."""
    rng = np.random.default_rng(8)
    rec = K.random_triangles(rng, 300_000)
    a = oracle.kat_a(capi.KAT_TRIANGLE_HIT, rec)
    b64, b32 = oracle.kat_b(capi.KAT_TRIANGLE_HIT, rec, F64), oracle.kat_b(capi.KAT_TRIANGLE_HIT, rec, F32)
    assert (a[:, 0] != b64[:, 0]).sum() <= 3
    assert (a[:, 0] != b32[:, 0]).mean() < 2e-5
    both = (a[:, 0] == 1) & (b32[:, 0] == 1)
    assert 0.2 < both.mean() < 0.8
    rel = np.abs(a[both, 1] - b32[both, 1]) / np.abs(a[both, 1])  # grazing rays (det -> 0) amplify without bound
    assert np.median(rel) < 2e-7 and np.quantile(rel, 0.999) < 1e-4


def test_scan_block_filter_is_conservative_against_the_reference_discriminant(oracle):
    """RAYZ_KAT_SCAN_DISCS on the CPU: mode B's reject test (r_pad^2 - p1^2 - p2^2, f32, the flat list's and the leaves') never
    rejects a sphere whose line the reference's own discriminant (src/geom.zig:40-50, mode A, f64) says is met — the
    r = 1000 ground sphere included — and it rejects the clear misses (it is a filter, not a pass-through)."""
    rng = np.random.default_rng(31)
    rec = K.random_scan_blocks(rng, 300_000)
    a = oracle.kat_a(capi.KAT_SCAN_DISCS, rec)[:, :4]
    for prec in (F32, F64):
        b = oracle.kat_b(capi.KAT_SCAN_DISCS, rec, prec)
        assert (b[:, :4] == b[:, 4:8]).all()  # mode B has ONE form of the test
        hit = a >= 0
        assert 0.1 < hit.mean() < 0.6
        assert (b[:, :4][hit] >= 0).all(), int((b[:, :4][hit] < 0).sum())
        # relative discriminant: a^-1 |oc|^-2-scaled; clear misses (the line passes further than 1.05 r + 1e-3 |oc| away) are filtered out
        c = np.stack([rec[:, 0:4], rec[:, 4:8] + rec[:, 16:20] * rec[:, 26:27] * rec[:, 27:28], rec[:, 8:12]], 2)
        oc = c - rec[:, None, 20:23]
        dd = K.unit(rec[:, 23:26])[:, None, :]
        dist = np.linalg.norm(oc - (oc * dd).sum(2, keepdims=True) * dd, axis=2)
        clear = dist > 1.05 * rec[:, 12:16] + 1e-3 * np.linalg.norm(oc, axis=2)
        assert clear.mean() > 0.3 and (b[:, :4][clear] < 0).all()
