"""Pin the oracle against every known-answer vector the reference's own tests hold,
plus the published vectors of the third-party generators it restates."""
import ctypes as C

import numpy as np
import pytest

from rayz_amd import capi


def d3(v):
    return capi.D3(*[float(x) for x in v])


def v3op(lib, op, a, b=None):
    out = capi.D3()
    lib.rayz_oracle_v3_op(op, d3(a), d3(b) if b is not None else None, out)
    return np.array(out)


# ---- src/vec.zig:169-215 -------------------------------------------------------------------------
def test_v3_add(oracle):  # "v3 add"
    lib = oracle.load()
    assert lib.rayz_oracle_v3_mag(d3([0, 0, 1])) == 1
    assert v3op(lib, 0, [0, 0, 1], [-1, 1, 0]).tolist() == [-1, 1, 1]


def test_v3_mul(oracle):  # "v3 mul"
    lib = oracle.load()
    assert v3op(lib, 2, [-1, 1, 0], [-2.5, 0, 0]).tolist() == [2.5, -2.5, 0]


def test_v3_dot_mag_unit(oracle):  # "v3 dot+mag+unit"
    lib = oracle.load()
    a, b = [0, 1, 0], [1, 0, 0]
    assert lib.rayz_oracle_v3_dot(d3(a), d3(b)) == 0
    assert lib.rayz_oracle_v3_dot(d3(a), d3(a)) == 1
    assert lib.rayz_oracle_v3_dot(d3(v3op(lib, 2, a, [2, 0, 0])), d3(a)) == 2
    assert lib.rayz_oracle_v3_dot(d3(a), d3([0.5, 0.5, 1])) == 0.5
    c = [4.5, -1.2, 3.3]
    assert lib.rayz_oracle_v3_dot(d3(c), d3(c)) == 32.58
    assert lib.rayz_oracle_v3_mag(d3(c)) == pytest.approx(5.7078, rel=1e-4)
    assert lib.rayz_oracle_v3_mag(d3(v3op(lib, 3, c))) == pytest.approx(1, rel=1e-4)
    assert lib.rayz_oracle_v3_mag(d3(v3op(lib, 3, v3op(lib, 0, a, b)))) == pytest.approx(1, rel=1e-4)


def test_v3_amax(oracle):  # "amax"
    lib = oracle.load()
    assert lib.rayz_oracle_v3_amax(d3([10, 2, 0])) == 0
    assert lib.rayz_oracle_v3_amax(d3([-1, 2, 0])) == 1
    assert lib.rayz_oracle_v3_amax(d3([-1, 2, 3])) == 2


def test_v3_div_is_reciprocal_multiply(oracle):  # src/vec.zig:67-69
    lib = oracle.load()
    v = [1.0, 7.0, 0.1]
    assert v3op(lib, 8, v, [3, 0, 0]).tolist() == [x * (1 / 3.0) for x in v]


# ---- src/utils.zig:15-32 (clamp is the only one the path uses: src/vec.zig:79-85) ---------------
def test_clamp(oracle):
    lib = oracle.load()
    assert v3op(lib, 7, [0.01, -2.999, 2.999], [0.0, 1.0, 0]).tolist() == [0.01, 0.0, 1.0]
    assert v3op(lib, 7, [-2.999, 2.999, -0.5], [-1.0, 0.0, 0]).tolist() == [-1.0, 0.0, -0.5]


# ---- src/geom.zig:69-84 ----------------------------------------------------------------------------
def test_sphere_bbox(oracle):  # "sphere bbox"
    lib = oracle.load()
    lo, hi = capi.D3(), capi.D3()
    s = capi.Sphere(center=d3([0, 0, 0]), velocity=d3([0, 0, 0]), radius=1.0, material=0)
    lib.rayz_oracle_sphere_bbox(C.byref(s), lo, hi)
    assert list(lo) == [-1, -1, -1] and list(hi) == [1, 1, 1]
    m = capi.Sphere(center=d3([0, 0, 0]), velocity=d3([1, 1, 1]), radius=1.0, material=0)
    lib.rayz_oracle_sphere_bbox(C.byref(m), lo, hi)
    assert list(lo) == [-1, -1, -1] and list(hi) == [2, 2, 2]


# ---- src/hit.zig:237-279 ----------------------------------------------------------------------------
def test_enclose_bbox(oracle):  # "enclose bbox"
    lib = oracle.load()
    lo, hi = capi.D3(), capi.D3()
    lib.rayz_oracle_aabb_enclose(d3([1] * 3), d3([-1] * 3), d3([0] * 3), d3([2] * 3), lo, hi)
    assert list(lo) == [-1, -1, -1] and list(hi) == [2, 2, 2]


def test_bbox_hit(oracle):  # "bbox hit"
    lib = oracle.load()
    box = (d3([0] * 3), d3([1] * 3))
    assert lib.rayz_oracle_aabb_hit(*box, d3([-1] * 3), d3([1] * 3), 0, 10) == 1
    assert lib.rayz_oracle_aabb_hit(*box, d3([-1] * 3), d3([-1] * 3), 0, 10) == 0
    assert lib.rayz_oracle_aabb_hit(*box, d3([-1] * 3), d3([0.5] * 3), 0, 10) == 1


def test_bbox_hit_2(oracle):  # "bbox hit 2": the real camera ray against the ground's box
    lib = oracle.load()
    assert lib.rayz_oracle_aabb_hit(d3([-1000, -2000, -1000]), d3([1000, 2, 1000]), d3([13, 2, 3]),
                                    d3([-9.6, -1.5, -2.3]), 0, 10) == 1


# ---- src/material.zig:213-223 -----------------------------------------------------------------------
def test_refract(oracle):  # "refract"
    lib = oracle.load()
    a = np.array([-0.3125, -0.3125, -1.0])
    a /= np.sqrt((a * a).sum())
    out = capi.D3()
    lib.rayz_oracle_refract(d3(a), d3([-0.558127, -0.558127, 0.613994]), 1.0 / 1.5, out)
    assert out[0] == pytest.approx(0.144881, rel=1e-4)
    assert out[1] == pytest.approx(0.144881, rel=1e-4)
    assert out[2] == pytest.approx(-0.978784, rel=1e-4)


# ---- src/renderer.zig:129-149 ("get ray"; the test is stale — it passes 6 arguments to an 8-argument
#      Camera.init — and its numbers hold for focus_dist = |from - at| = sqrt(12), defocus_angle = 0) --
def test_get_ray(oracle):
    lib = oracle.load()
    cam = capi.CameraDesc()
    lib.rayz_oracle_camera_init(90, 12 ** 0.5, 0, d3([-2, 2, 1]), d3([0, 0, -1]), d3([0, 1, 0]), 225, 400, cam)
    o, d = capi.D3(), capi.D3()
    lib.rayz_oracle_get_ray_norng(cam, 0, 0, o, d)
    assert list(o) == [-2, 2, 1]
    for got, want in zip(d, (-0.935834, 0.815856, -7.75169)):
        assert got == pytest.approx(want, rel=1e-5)
    lib.rayz_oracle_get_ray_norng(cam, 112, 199, o, d)
    for got, want in zip(d, (-0.998817, -4.18732, -2.8115)):
        assert got == pytest.approx(want, rel=1e-5)


def test_get_ray_without_a_generator_through_the_kat_records(oracle):
    """The same vector through the known-answer record format with n_u = -1 (= `getRay(px, py, null)`, the call the reference's test
    makes): mode A (the reference's function with rng == null) and mode B (the kernel arithmetic, what rayz_hip_kat's device code is
    held to bit for bit in tests/test_kat_gpu.py) both reproduce it; no draw is consumed and time is 0."""
    import kat_records as K

    cam = capi.CameraDesc()
    oracle.load().rayz_oracle_camera_init(90, 12 ** 0.5, 0, d3([-2, 2, 1]), d3([0, 0, -1]), d3([0, 1, 0]), 225, 400, cam)
    rec, want = K.get_ray_reference(cam, no_rng=True)
    assert rec[:, 21].tolist() == [-1, -1]
    for got in (oracle.kat_a(capi.KAT_GET_RAY, rec), oracle.kat_b(capi.KAT_GET_RAY, rec, capi.PRECISION_F64),
                oracle.kat_b(capi.KAT_GET_RAY, rec, capi.PRECISION_F32)):
        assert got[:, 0:3].tolist() == [[-2, 2, 1]] * 2 and got[:, 3:6] == pytest.approx(want, rel=1e-5)
        assert got[:, 6].tolist() == [0, 0] and got[:, 7].tolist() == [0, 0]  # time 0, no draw
    jit, _ = K.get_ray_reference(cam)  # n_u = 0: every draw 0.5 — the same ray (x + 0 exactly), but three draws and time 0.5
    a0, a5 = oracle.kat_a(capi.KAT_GET_RAY, rec), oracle.kat_a(capi.KAT_GET_RAY, jit)
    assert np.array_equal(a0[:, 0:6], a5[:, 0:6]) and a5[:, 6].tolist() == [0.5, 0.5] and a5[:, 7].tolist() == [3, 3]


# ---- src/image.zig:32-39 + src/vec.zig:79-93: sqrt (0 for non-positive), clamp, truncating *255 -----
@pytest.mark.parametrize("rgb,want", [
    ([0.0, 1.0, 4.0], [0, 255, 255]),
    ([-1.0, 0.25, 0.5], [0, 127, 180]),
    ([1e-6, 0.9999, 0.04], [0, 254, 51]),
])
def test_ppm_transform(oracle, rgb, want):
    lib = oracle.load()
    out = (C.c_uint8 * 3)()
    lib.rayz_oracle_ppm_u8(d3(rgb), out)
    assert list(out) == want


# ---- third-party generators (Zig std is not under /root/reference): published known answers --------
def test_splitmix64_published_vector(oracle):
    """Vigna's splitmix64, seed 1234567 (the widely published first five outputs)."""
    lib = oracle.load()
    out = (C.c_uint64 * 5)()
    lib.rayz_oracle_splitmix64(1234567, 5, out)
    assert list(out) == [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431,
                         16408922859458223821]


def test_pcg32_published_vector(oracle):
    """O'Neill's pcg32-demo: pcg32_srandom(42, 54) -> these six outputs."""
    lib = oracle.load()
    out = (C.c_uint32 * 6)()
    lib.rayz_oracle_pcg32(42, 54, 6, out)
    assert list(out) == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]


def test_xoshiro_float_in_unit_interval_and_deterministic(oracle):
    """DefaultPrng (xoshiro256++ via SplitMix64): no published vector is held by the reference, so this only
    pins self-consistency (parity unpinned for the Zig std stream; it cannot matter — the reference's seed is
    unobservable)."""
    lib = oracle.load()
    s1, s2 = (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
    lib.rayz_oracle_xoshiro_seed(99, s1)
    lib.rayz_oracle_xoshiro_seed(99, s2)
    a, b = (C.c_double * 1000)(), (C.c_double * 1000)()
    lib.rayz_oracle_xoshiro_f64(s1, 1000, a)
    lib.rayz_oracle_xoshiro_f64(s2, 1000, b)
    a = np.array(a)
    assert (a == np.array(b)).all() and (a >= 0).all() and (a < 1).all()
    assert 0.45 < a.mean() < 0.55
    # state seeding = four splitmix64 outputs
    sm = (C.c_uint64 * 4)()
    lib.rayz_oracle_splitmix64(99, 4, sm)
    s3 = (C.c_uint64 * 4)()
    lib.rayz_oracle_xoshiro_seed(99, s3)
    assert list(s3) == list(sm)
