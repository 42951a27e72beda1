"""The host-side mirror of the reference's Tracer / MemPool / Camera / Image API (rayz_amd/host/rayz.hpp
through include/rayz_host.h), checked against the reference's own vectors and against the oracle's
independent restatement of the same host code.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from rayz_amd import capi, tracer


def test_tracer_defaults_match_reference(built):
    """src/renderer.zig:23-24 (max_bounces 50, samples_per_px 10) and :39-40 (h = floor(w / (16/9)))."""
    for w, h in [(400, 225), (1920, 1080), (3840, 2160), (100, 56), (401, 225)]:
        t = tracer.Tracer.init(w, 20.0, 10.0, 0.6, (13, 2, 3), (0, 0, 0), (0, 1, 0), seed=1)
        i = t.info()
        assert (i.width, i.height) == (w, h)
        assert (i.samples_per_px, i.max_bounces) == (10, 50)
        p = t.params()
        assert p.tmin == 1e-3 and p.precision == capi.PRECISION_F32 and p.traversal == capi.TRAVERSAL_AUTO
    t.set_gpu(precision=capi.PRECISION_F64)
    assert t.params().tmin == 1e-10  # the reference's own tmin, src/renderer.zig:107


def test_get_ray_reference_vector(built):
    """The reference's "get ray" test (src/renderer.zig:129-149) through the product's Camera."""
    t = tracer.Tracer.init(400, 90.0, 12 ** 0.5, 0.0, (-2, 2, 1), (0, 0, -1), (0, 1, 0), seed=1)
    o, d = t.get_ray(0, 0)
    assert o.tolist() == [-2, 2, 1]
    assert d == pytest.approx([-0.935834, 0.815856, -7.75169], rel=1e-5)
    _, d = t.get_ray(112, 199)
    assert d == pytest.approx([-0.998817, -4.18732, -2.8115], rel=1e-5)


def test_camera_init_equals_oracle(built, oracle):
    lib = oracle.load()
    for args in [(20.0, 10.0, 0.6, (13, 2, 3), (0, 0, 0), (0, 1, 0), 400),
                 (90.0, 3.4, 10.0, (-2, 2, 1), (0, 0, -1), (0, 1, 0), 1920),
                 (35.0, 1.0, 0.0, (0, 0, 5), (1, 1, 0), (0.1, 1, 0), 123)]:
        vfov, fd, da, lf, la, up, w = args
        t = tracer.Tracer.init(w, vfov, fd, da, lf, la, up, seed=1)
        want = capi.CameraDesc()
        lib.rayz_oracle_camera_init(vfov, fd, da, capi.D3(*lf), capi.D3(*la), capi.D3(*up), t.info().height, w, want)
        assert bytes(t.camera_desc()) == bytes(want)


def test_pool_handles_are_insertion_indices(built):
    """MemPool.addAndReturnHandle, src/ecs.zig:57-69."""
    t = tracer.Tracer.init(64, 20, 1, 0, (0, 0, 1), (0, 0, 0), (0, 1, 0), seed=3)
    a = t.pool.add_solid_texture((0.1, 0.2, 0.3))
    b = t.pool.add_solid_texture((0.9, 0.9, 0.9))
    c = t.pool.add_checker_texture(0.5, a, b)
    assert (a, b, c) == (0, 1, 2)
    m0 = t.pool.add_diffuse(c)
    m1 = t.pool.add_metallic(a, fuzz=0.25)
    m2 = t.pool.add_dielectric(1.5)
    assert (m0, m1, m2) == (0, 1, 2)
    s0 = t.pool.add_sphere((0, 0, -1), 0.5, m0)
    s1 = t.pool.add_sphere((1, 0, -1), 0.25, m1, velocity=(0, 0.5, 0))
    assert (s0, s1) == (0, 1)
    sd = t.scene_desc()
    assert (sd.n_spheres, sd.n_materials, sd.n_textures) == (2, 3, 3)
    assert sd.textures[2].kind == capi.TEX_CHECKER and (sd.textures[2].even, sd.textures[2].odd) == (0, 1)
    assert sd.textures[2].scale == 0.5 and list(sd.textures[0].color) == [0.1, 0.2, 0.3]
    assert sd.materials[0].kind == capi.MAT_DIFFUSE and sd.materials[0].method == capi.DIFFUSE_HEMISPHERE
    assert sd.materials[1].kind == capi.MAT_METALLIC and sd.materials[1].param == 0.25
    assert sd.materials[2].kind == capi.MAT_DIELECTRIC and sd.materials[2].param == 1.5
    assert list(sd.spheres[1].velocity) == [0, 0.5, 0] and sd.spheres[1].radius == 0.25
    assert sd.spheres[1].material == 1


@pytest.mark.parametrize("seed,lo,hi", [(42, -11, 11), (7, -11, 11), (1, -3, 3), (5, -50, 50)])
def test_random_bouncing_equals_oracle_restatement(built, oracle, seed, lo, hi):
    """Product scene generator vs the oracle's independent restatement of src/rayz.zig:45-168: identical
    pools, draw for draw, and the Tracer's stream left in the same state (src/rayz.zig:109)."""
    t = tracer.randomBouncing(400, lo, hi, seed=seed)
    o = oracle.OracleScene(seed, lo, hi)
    sd = t.scene_desc()
    assert (sd.n_spheres, sd.n_materials, sd.n_textures) == o.counts
    for name, T, n in [("spheres", capi.Sphere, o.counts[0]), ("materials", capi.Material, o.counts[1]),
                       ("textures", capi.Texture, o.counts[2])]:
        got = C.string_at(getattr(sd, name), C.sizeof(T) * n)
        want = C.string_at(getattr(o, name), C.sizeof(T) * n)
        assert got == want, name
    assert (t.rng_state() == o.rng_state).all()
    if (lo, hi) == (-11, 11):
        assert sd.n_spheres <= 488  # 22*22 + 4 minus rejects
        # 80 / 15 / 5 % split, diffuse ones moving in +y by [0, 0.5)
        kinds = np.array([sd.materials[sd.spheres[i].material].kind for i in range(4, sd.n_spheres)])
        vy = np.array([sd.spheres[i].velocity[1] for i in range(4, sd.n_spheres)])
        assert 0.7 < (kinds == capi.MAT_DIFFUSE).mean() < 0.9
        assert ((vy > 0) == (kinds == capi.MAT_DIFFUSE)).all() and (vy < 0.5).all()


def test_random_bouncing_10k_scene_shape(built):
    t = tracer.randomBouncing(1920, -50, 50, seed=42)
    i = t.info()
    assert (i.width, i.height) == (1920, 1080)
    assert 9990 <= i.n_spheres <= 10004
    sd = t.scene_desc()
    assert sd.spheres[0].radius == 1000 and list(sd.spheres[0].center) == [0, -1000, 0]
    assert sd.materials[sd.spheres[0].material].texture == 2 and sd.textures[2].kind == capi.TEX_CHECKER


def test_rng_mirror_equals_oracle(built, oracle):
    lib = oracle.load()
    t = tracer.Tracer.init(64, 20, 1, 0, (0, 0, 1), (0, 0, 0), (0, 1, 0), seed=123456789)
    st = (C.c_uint64 * 4)()
    lib.rayz_oracle_xoshiro_seed(123456789, st)
    assert t.rng_state().tolist() == list(st)
    want_u = (C.c_uint64 * 8)()
    lib.rayz_oracle_xoshiro_u64(st, 8, want_u)
    assert [t.rng_next() for _ in range(8)] == list(want_u)
    want_f = (C.c_double * 8)()
    lib.rayz_oracle_xoshiro_f64(st, 8, want_f)
    assert [t.rng_float() for _ in range(8)] == list(want_f)


def test_unseeded_tracers_differ(built):
    """The reference seeds from the OS (src/renderer.zig:55-59); so does the mirror when no seed is given."""
    a = tracer.Tracer.init(64, 20, 1, 0, (0, 0, 1), (0, 0, 0), (0, 1, 0))
    b = tracer.Tracer.init(64, 20, 1, 0, (0, 0, 1), (0, 0, 0), (0, 1, 0))
    assert a.rng_state().tolist() != b.rng_state().tolist()


def test_write_ppm_equals_oracle_writer(built, oracle, tmp_path):
    """Image.writePPM, src/image.zig:29-41: header, one "r g b" line per pixel, sqrt-gamma, clamp, truncation."""
    rng = np.random.default_rng(0)
    img = tracer.Image(9, 16)
    img.pixels = rng.uniform(-0.2, 1.5, size=(9, 16, 3))
    img.pixels[0, 0] = [0.0, 1.0, 0.25]
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    img.writePPM(a)
    px = np.ascontiguousarray(img.pixels)
    assert oracle.load().rayz_oracle_write_ppm(b.encode(), px.ctypes.data_as(C.POINTER(C.c_double)), 16, 9) == 0
    ta, tb = open(a).read(), open(b).read()
    assert ta == tb
    lines = ta.split("\n")
    assert lines[:3] == ["P3", "16 9", "255"] and lines[3] == "0 255 127" and len(lines) == 3 + 144 + 1
    assert (img.to_u8().reshape(-1, 3)[0] == [0, 255, 127]).all()


def test_cli_driver_usage(built):
    """`rayz` without arguments: the reference panics on the missing img_w (src/rayz.zig:16); the mirror exits 2."""
    import subprocess

    exe = os.path.join(os.path.dirname(tracer.__file__), "host", "rayz")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([exe, "abc"], capture_output=True, text=True)
    assert r.returncode == 1 and "InvalidCharacter" in r.stderr
