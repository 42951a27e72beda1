"""The HIP device functions themselves — `rayz_hip_kat` runs the trace kernels' own inlined functions, one GPU thread
per record — held (1) to the reference's test vectors directly and (2) to the oracle's mode B bit for bit on random
records of every op, f32 and f64.  Together with tests/test_kat_cpu.py (mode B == mode A deterministically) this
closes the chain device code -> mode B -> the reference's functions without a statistical link."""
import numpy as np
import pytest

import kat_records as K
from rayz_amd import capi

pytestmark = pytest.mark.gpu
F32, F64 = capi.PRECISION_F32, capi.PRECISION_F64


@pytest.mark.parametrize("prec", [F32, F64])
def test_device_functions_on_the_reference_vectors(gpu, oracle, prec):
    rec, want = K.refract_reference()  # src/material.zig:213-223
    assert gpu.kat(capi.KAT_REFRACT, rec, prec)[0, :3] == pytest.approx(want, rel=1e-4)
    cam = capi.CameraDesc()
    oracle.load().rayz_oracle_camera_init(90, 12 ** 0.5, 0, oracle.d3([-2, 2, 1]), oracle.d3([0, 0, -1]),
                                          oracle.d3([0, 1, 0]), 225, 400, cam)
    rec, want = K.get_ray_reference(cam)  # src/renderer.zig:129-149
    got = gpu.kat(capi.KAT_GET_RAY, rec, prec)
    assert got[:, 0:3].tolist() == [[-2, 2, 1]] * 2 and got[:, 3:6] == pytest.approx(want, rel=1e-5)
    # .. and the reference's own call, getRay(px, py, null) (src/camera.zig:59-77 with rng == null; n_u = -1): no draw, time 0,
    # the same ray as the all-0.5 jitter gives — bit for bit — and bit-identical to mode B; mode A agrees to f64 / f32 rounding
    rec0, _ = K.get_ray_reference(cam, no_rng=True)
    got0 = gpu.kat(capi.KAT_GET_RAY, rec0, prec)
    assert got0[:, 7].tolist() == [0, 0] and got0[:, 6].tolist() == [0, 0] and got[:, 7].tolist() == [3, 3]  # draws made, time
    assert np.array_equal(got0[:, 0:6], got[:, 0:6]) and got0[:, 3:6] == pytest.approx(want, rel=1e-5)
    assert np.array_equal(got0, oracle.kat_b(capi.KAT_GET_RAY, rec0, prec))
    assert oracle.kat_a(capi.KAT_GET_RAY, rec0)[:, 3:6] == pytest.approx(want, rel=1e-5)
    rec, want = K.box_hit_reference()  # src/hit.zig:247-279
    assert gpu.kat(capi.KAT_BOX_HIT, rec, prec)[:, 0].tolist() == want.tolist()


@pytest.mark.parametrize("prec", [F32, F64])
def test_device_functions_equal_mode_b_bit_for_bit(gpu, oracle, prec):
    rng = np.random.default_rng(77)
    cam = capi.CameraDesc()
    oracle.load().rayz_oracle_camera_init(20, 10.0, 0.6, oracle.d3([13, 2, 3]), oracle.d3([0, 0, 0]),
                                          oracle.d3([0, 1, 0]), 1080, 1920, cam)
    refl = K.blank(20_000)
    refl[:, 0], refl[:, 1] = rng.uniform(0, 1, 20_000), rng.uniform(0.4, 2.5, 20_000)
    bg = K.blank(20_000)
    bg[:, 0:3] = rng.normal(size=(20_000, 3))
    cases = [
        (capi.KAT_REFRACT, K.random_refracts(rng, 50_000)),
        (capi.KAT_REFLECTANCE, refl),
        (capi.KAT_GET_RAY, K.random_get_rays(rng, 50_000, cam)),
        (capi.KAT_BOX_HIT, K.random_boxes(rng, 100_000)),
        (capi.KAT_BOX_HIT, K.axis_parallel_boxes(rng, 50_000)[0]),  # direction components of exactly ±0
        (capi.KAT_SPHERE_HIT, K.random_sphere_hits(rng, 100_000)),
        (capi.KAT_SPHERE_HIT, K.random_sphere_hits(rng, 100_000, big=True)),
        (capi.KAT_SCATTER, K.random_scatters(rng, 100_000)),
        (capi.KAT_CHECKER, K.random_checkers(rng, 50_000)),
        (capi.KAT_BACKGROUND, bg),
        (capi.KAT_TRIANGLE_HIT, K.random_triangles(rng, 100_000)),
        (capi.KAT_SCAN_DISCS, K.random_scan_blocks(rng, 200_000)),  # the flat list's PACKED-FMA reject test vs mode B's scalar one
    ]
    for op, rec in cases:
        got, want = gpu.kat(op, rec, prec), oracle.kat_b(op, rec, prec)
        same = (got == want) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (op, prec, int((~same).any(1).sum()), np.flatnonzero((~same).any(1))[:5].tolist())


@pytest.mark.parametrize("prec", [F32, F64])
def test_device_box_test_with_zero_direction_components(gpu, prec):
    """The device's slab test never culls a box whose slab the axis-parallel ray lies in (see the CPU test of the same
    name for the history)."""
    rec, truth, beside = K.axis_parallel_boxes(np.random.default_rng(12), 200_000)
    got = gpu.kat(capi.KAT_BOX_HIT, rec, prec)[:, 0]
    assert (got[truth] == 1).all() and (got[beside] == 0).all()


def test_conservative_filter_on_the_device(gpu, oracle):
    """Grazing rays on the r = 1000 ground sphere, the regime where the f32 filter used to be able to drop a sphere
    the f64 quadratic hits: whenever the reference's hitInner (mode A) hits, the device filter
    must have passed the pair on."""
    rng = np.random.default_rng(4)
    n = 200_000
    rec = K.blank(n)
    c, r = np.array([0.0, -1000.0, 0.0]), 1000.0
    o = c + K.unit(rng.normal(size=(n, 3)) + [0, 2, 0]) * (r * (1 + 10.0 ** rng.uniform(-7, -2, n)))[:, None]
    # the tangent cone from o makes the angle asin(r / |o - c|) with the direction to the centre: aim along it, a hair
    # inside or outside
    up = K.unit(o - c)
    tang = K.unit(np.cross(up, rng.normal(size=(n, 3))))
    alpha = np.arcsin(r / np.linalg.norm(o - c, axis=1)) + rng.normal(size=n) * 10.0 ** rng.uniform(-9, -2, n)
    d = (tang * np.sin(alpha)[:, None] - up * np.cos(alpha)[:, None]) * rng.uniform(0.5, 2.0, (n, 1))
    rec[:, 0:3], rec[:, 6] = c, r
    rec[:, 7:10], rec[:, 10:13] = K.f32r(o), K.f32r(d)
    rec[:, 13], rec[:, 14], rec[:, 15] = 0.5, 1e-3, np.inf
    a = oracle.kat_a(capi.KAT_SPHERE_HIT, rec)
    g = gpu.kat(capi.KAT_SPHERE_HIT, rec, F32)
    hit = a[:, 0] == 1
    assert 0.2 < hit.mean() < 0.8  # the set straddles the tangent
    assert (g[hit, 9] == 1).all(), int((g[hit, 9] == 0).sum())
    # and the padding is thin (E = 7.6e-3 on this sphere): a line that misses by more than 2 E is filtered out
    q, dd = c - rec[:, 7:10], K.unit(rec[:, 10:13])
    miss = np.linalg.norm(q - (q * dd).sum(1, keepdims=True) * dd, axis=1) - r
    assert (miss > 0.016).sum() > 1000 and (g[miss > 0.016, 9] == 0).all()


def test_kat_refuses_a_uniform_list_that_leaves_the_record(gpu):
    """n_u is the caller's: a list that would run past the record's 48 doubles (the device reads u[0 .. n_u)) is refused on
    the host with RAYZ_ERR_BAD_ARG, as is a negative (but for GET_RAY's -1), fractional or NaN count."""
    for op, at, cap in ((capi.KAT_GET_RAY, 21, 26), (capi.KAT_SCATTER, 16, 31)):
        rec = K.blank(3)
        rec[:, at] = cap
        gpu.kat(op, rec)  # the largest list that fits
        # (-1 is GET_RAY's "no generator" = getRay(px, py, null), accepted there and only there)
        for bad in (cap + 1, -2, -1.5, 2.5, float("nan"), 1e30) + (() if op == capi.KAT_GET_RAY else (-1,)):
            rec = K.blank(3)
            rec[2, at] = bad
            with pytest.raises(capi.RayzHipError, match="n_u"):
                gpu.kat(op, rec)
    rec = K.blank(3)
    rec[:, 21] = -1
    assert gpu.kat(capi.KAT_GET_RAY, rec)[:, 7].tolist() == [0, 0, 0]  # no draw made


@pytest.mark.parametrize("prec", [F32, F64])
def test_packed_scan_test_equals_the_leaf_form_on_the_device(gpu, prec):
    """ScanGroup::discs (two spheres per v_pk_fma_f32, what the flat list's scan loop runs) and the general-velocity form
    of the BVH leaves give the SAME value, bit for bit, for static and y-moving spheres: flat list and BVH filter alike."""
    rec = K.random_scan_blocks(np.random.default_rng(8), 200_000)
    got = gpu.kat(capi.KAT_SCAN_DISCS, rec, prec)
    assert (got[:, :4] == got[:, 4:8]).all() and np.isfinite(got[:, :8]).all()
    assert 0.1 < (got[:, :4] >= 0).mean() < 0.6
