"""The C-ABI library loads on a machine without a GPU and exports exactly what include/*.h declares;
argument checking and error reporting work without touching a device.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rayz_amd import capi, render, tracer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rayz_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    lib = capi.load()
    names = declared_symbols("rayz_hip.h") + declared_symbols("rayz_host.h")
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/ but not exported by librayz_hip.so"


def test_binding_covers_every_declared_symbol(built):
    bound = {p[0] for p in capi.PROTOTYPES} | {p[0] for p in tracer.HOST_PROTOTYPES}
    declared = set(declared_symbols("rayz_hip.h") + declared_symbols("rayz_host.h"))
    assert declared == bound, (declared - bound, bound - declared)


def test_abi_version_and_struct_sizes(built):
    lib = capi.load()
    assert lib.rayz_hip_abi_version() == capi.ABI_VERSION == 5
    # sizes the Zig extern structs must reproduce
    assert (C.sizeof(capi.Texture), C.sizeof(capi.Material), C.sizeof(capi.Sphere)) == (48, 24, 64)
    assert (C.sizeof(capi.SceneDesc), C.sizeof(capi.CameraDesc), C.sizeof(capi.Triangle)) == (48, 152, 80)
    assert (C.sizeof(capi.RenderParams), C.sizeof(capi.RenderStats)) == (56, 40)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(capi.RayzHipError, match="no CPU fallback"):
        capi.load(str(tmp_path / "nope.so"))


@pytest.mark.parametrize("h,tile,count", [(1080, 8, 1), (1080, 8, 8), (2160, 8, 8), (225, 8, 2), (7, 8, 3), (1, 1, 1),
                                          (100, 16, 5)])
def test_shard_rows_partition_the_image(built, h, tile, count):
    lib = capi.load()
    seen = []
    for idx in range(count):
        p = capi.RenderParams(width=4, height=h, samples_per_px=1, tile_rows=tile, shard_index=idx, shard_count=count)
        rows = render.shard_row_indices(h, tile, idx, count)
        assert lib.rayz_hip_shard_rows(C.byref(p)) == len(rows)
        seen.extend(rows.tolist())
    assert sorted(seen) == list(range(h))


def test_shard_rows_rejects_bad_index(built):
    lib = capi.load()
    p = capi.RenderParams(width=4, height=10, samples_per_px=1, shard_index=3, shard_count=2)
    assert lib.rayz_hip_shard_rows(C.byref(p)) == 0


def test_scene_validation_without_a_device(built):
    """Bad handles are rejected with RAYZ_ERR_BAD_ARG and a message, before any HIP call."""
    lib = capi.load()
    t = tracer.threeSpheres(64, seed=1)
    sd = t.scene_desc()
    h = C.c_void_p()
    assert lib.rayz_hip_scene_create(C.byref(sd), C.byref(h)) == capi.OK
    assert lib.rayz_hip_scene_destroy(h) == capi.OK
    # sphere with a material handle out of range
    bad = (capi.Sphere * 1)(capi.Sphere(center=capi.D3(0, 0, 0), velocity=capi.D3(0, 0, 0), radius=1, material=7))
    sd2 = capi.SceneDesc(spheres=bad, materials=sd.materials, textures=sd.textures, n_spheres=1,
                         n_materials=sd.n_materials, n_textures=sd.n_textures)
    assert lib.rayz_hip_scene_create(C.byref(sd2), C.byref(h)) == capi.ERR_BAD_ARG
    assert b"material handle 7" in lib.rayz_hip_last_error()
    assert lib.rayz_hip_scene_create(None, C.byref(h)) == capi.ERR_BAD_ARG
    # checker texture pointing outside the list
    tex = (capi.Texture * 1)(capi.Texture(kind=capi.TEX_CHECKER, even=0, odd=5, scale=1.0))
    sd3 = capi.SceneDesc(spheres=None, materials=None, textures=tex, n_textures=1)
    assert lib.rayz_hip_scene_create(C.byref(sd3), C.byref(h)) == capi.ERR_BAD_ARG


def test_render_without_device_is_an_error_not_a_fallback(built):
    """On a box with no GPU the render entry points must fail (NO_DEVICE), never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the gpu tests")
    lib = capi.load()
    assert lib.rayz_hip_init(0) == capi.ERR_NO_DEVICE
    t = tracer.threeSpheres(32, seed=1)
    t.samples_per_px = 1
    out = np.full((18, 32, 3), -1.0, dtype=np.float32)
    rc = lib.rayz_hip_render(C.byref(t.scene_desc()), C.byref(t.camera_desc()), C.byref(t.params()),
                             out.ctypes.data_as(C.c_void_p), None)
    assert rc == capi.ERR_NO_DEVICE
    assert (out == -1.0).all()
    with pytest.raises(capi.RayzHipError):
        t.render()


def test_param_validation(built):
    lib = capi.load()
    t = tracer.threeSpheres(32, seed=1)
    out = np.zeros((18, 32, 3), dtype=np.float32)
    p = t.params()
    p.samples_per_px = 0
    rc = lib.rayz_hip_render(C.byref(t.scene_desc()), C.byref(t.camera_desc()), C.byref(p),
                             out.ctypes.data_as(C.c_void_p), None)
    assert rc == capi.ERR_BAD_ARG
    p = t.params()
    p.precision = 9
    rc = lib.rayz_hip_render(C.byref(t.scene_desc()), C.byref(t.camera_desc()), C.byref(p),
                             out.ctypes.data_as(C.c_void_p), None)
    assert rc == capi.ERR_BAD_ARG and b"precision" in lib.rayz_hip_last_error()


def _tex_scene(textures):
    arr = (capi.Texture * len(textures))(*textures)
    sd = capi.SceneDesc(spheres=None, materials=None, textures=arr, n_textures=len(textures))
    sd._keep = arr
    return sd


def test_checker_chains_the_device_loop_cannot_resolve_are_refused(built):
    """The reference recurses through nested checkers without a limit (src/material.zig:36-37); the device walks a
    bounded loop of 8 lookups.  A deeper chain or a cycle is RAYZ_ERR_BAD_ARG at scene creation, never a black pixel."""
    lib = capi.load()
    h = C.c_void_p()
    solid = capi.Texture(kind=capi.TEX_SOLID, color=capi.D3(1, 1, 1))

    def chain(n_checkers):  # texture 0 solid, texture k = checker(k-1, 0)
        return [solid] + [capi.Texture(kind=capi.TEX_CHECKER, even=k - 1, odd=0, scale=1.0) for k in range(1, n_checkers + 1)]

    sd = _tex_scene(chain(7))  # 7 checkers + the solid = 8 lookups: the limit
    assert lib.rayz_hip_scene_create(C.byref(sd), C.byref(h)) == capi.OK
    lib.rayz_hip_scene_destroy(h)
    sd = _tex_scene(chain(8))
    assert lib.rayz_hip_scene_create(C.byref(sd), C.byref(h)) == capi.ERR_BAD_ARG
    assert b"nesting depth 9" in lib.rayz_hip_last_error()
    # cycles: a checker naming itself, and a two-cycle
    sd = _tex_scene([solid, capi.Texture(kind=capi.TEX_CHECKER, even=1, odd=0, scale=1.0)])
    assert lib.rayz_hip_scene_create(C.byref(sd), C.byref(h)) == capi.ERR_BAD_ARG and b"cycle" in lib.rayz_hip_last_error()
    sd = _tex_scene([solid, capi.Texture(kind=capi.TEX_CHECKER, even=0, odd=2, scale=1.0),
                     capi.Texture(kind=capi.TEX_CHECKER, even=1, odd=0, scale=1.0)])
    assert lib.rayz_hip_scene_create(C.byref(sd), C.byref(h)) == capi.ERR_BAD_ARG and b"cycle" in lib.rayz_hip_last_error()
    # a diamond (two checkers sharing children) is fine
    sd = _tex_scene([solid, solid, capi.Texture(kind=capi.TEX_CHECKER, even=0, odd=1, scale=1.0),
                     capi.Texture(kind=capi.TEX_CHECKER, even=2, odd=2, scale=2.0),
                     capi.Texture(kind=capi.TEX_CHECKER, even=3, odd=2, scale=3.0)])
    assert lib.rayz_hip_scene_create(C.byref(sd), C.byref(h)) == capi.OK
    lib.rayz_hip_scene_destroy(h)


def test_multi_device_argument_validation(built):
    """rayz_hip_multi_create / rayz_hip_render_multi check their arguments before touching any device."""
    lib = capi.load()
    t = tracer.threeSpheres(32, seed=1)
    sd = t.scene_desc()
    h = C.c_void_p()

    def create(devs, transport=capi.GATHER_RCCL, scene=sd):
        arr = (C.c_int * max(len(devs), 1))(*devs)
        return lib.rayz_hip_multi_create(arr if devs is not None else None, len(devs), C.byref(scene) if scene else None,
                                         transport, C.byref(h))

    assert lib.rayz_hip_multi_create(None, 1, C.byref(sd), 0, C.byref(h)) == capi.ERR_BAD_ARG
    assert create([]) == capi.ERR_BAD_ARG and b"n_devices" in lib.rayz_hip_last_error()
    assert create(list(range(capi.MAX_DEVICES + 1))) == capi.ERR_BAD_ARG
    assert create([0, 1, 0]) == capi.ERR_BAD_ARG and b"twice" in lib.rayz_hip_last_error()
    assert create([-1]) == capi.ERR_BAD_ARG
    assert create([0], transport=7) == capi.ERR_BAD_ARG and b"transport" in lib.rayz_hip_last_error()
    assert create([0], scene=None) == capi.ERR_BAD_ARG
    arr = (C.c_int * 1)(0)
    assert lib.rayz_hip_multi_create(arr, 1, C.byref(sd), 0, None) == capi.ERR_BAD_ARG
    assert lib.rayz_hip_multi_render(None, C.byref(t.camera_desc()), C.byref(t.params()), None, None) == capi.ERR_STATE
    assert lib.rayz_hip_multi_destroy(None) == capi.OK
    import torch

    if not torch.cuda.is_available():  # no device here: a valid request must fail with NO_DEVICE, not compute
        assert create([0]) == capi.ERR_NO_DEVICE
        out = np.full((18, 32, 3), -1.0, dtype=np.float32)
        rc = lib.rayz_hip_render_multi(arr, 1, C.byref(sd), C.byref(t.camera_desc()), C.byref(t.params()),
                                       out.ctypes.data_as(C.c_void_p), None)
        assert rc == capi.ERR_NO_DEVICE and (out == -1.0).all()
        assert lib.rayz_hip_scene_create_on(0, C.byref(sd), C.byref(h)) == capi.ERR_NO_DEVICE
        t.set_gpu(devices=[0])
        with pytest.raises(capi.RayzHipError):
            t.render()


def test_errors_are_reported_per_thread(built):
    """rayz_hip_last_error() is thread-local: two threads working on two scenes see their own messages."""
    import threading

    lib = capi.load()
    t = tracer.threeSpheres(32, seed=1)
    sd = t.scene_desc()
    seen, barrier = {}, threading.Barrier(2)

    def work(name, bad_material):
        bad = (capi.Sphere * 1)(capi.Sphere(center=capi.D3(0, 0, 0), velocity=capi.D3(0, 0, 0), radius=1, material=bad_material))
        sd2 = capi.SceneDesc(spheres=bad, materials=sd.materials, textures=sd.textures, n_spheres=1,
                             n_materials=sd.n_materials, n_textures=sd.n_textures)
        h = C.c_void_p()
        for _ in range(200):
            rc = lib.rayz_hip_scene_create(C.byref(sd2), C.byref(h))
            barrier.wait()  # both threads have failed before either reads its message
            msg = lib.rayz_hip_last_error()
            if rc != capi.ERR_BAD_ARG or f"material handle {bad_material}".encode() not in msg:
                seen[name] = (rc, msg)
                barrier.abort()
                return
            try:
                barrier.wait()
            except threading.BrokenBarrierError:
                return
        seen[name] = "ok"

    ths = [threading.Thread(target=work, args=("a", 111)), threading.Thread(target=work, args=("b", 222))]
    [x.start() for x in ths]
    [x.join() for x in ths]
    assert seen == {"a": "ok", "b": "ok"}, seen


def test_chunk_schedule_properties_and_oracle_agreement(built, oracle):
    """The chunk schedule is part of the image's definition: product and oracle restate it
    independently and must agree; it partitions [0, spp), does not depend on the shard, keeps small renders on uniform
    chunks of 16 (the golden fixtures), and ends large renders in short chunks."""
    lib, olib = capi.load(), oracle.load()

    def sched(fn, **kw):
        p = capi.RenderParams(**kw)
        buf = (C.c_uint32 * 4200)()
        n = fn(C.byref(p), buf, 4200)
        return list(buf[: n + 1])

    cases = [dict(width=64, height=36, samples_per_px=8), dict(width=1920, height=1080, samples_per_px=1024),
             dict(width=1920, height=1080, samples_per_px=256), dict(width=3840, height=2160, samples_per_px=4096),
             dict(width=1920, height=1080, samples_per_px=1000), dict(width=1920, height=1080, samples_per_px=63),
             dict(width=1920, height=1080, samples_per_px=64), dict(width=1920, height=1080, samples_per_px=100, chunk_spp=7),
             dict(width=1920, height=1080, samples_per_px=1), dict(width=724, height=724, samples_per_px=77),
             dict(width=725, height=724, samples_per_px=77), dict(width=1920, height=1080, samples_per_px=5, chunk_spp=16)]
    for kw in cases:
        a, b = sched(lib.rayz_hip_chunk_schedule, **kw), sched(olib.rayz_oracle_chunk_schedule, **kw)
        spp = kw["samples_per_px"]
        assert a == b and a[0] == 0 and a[-1] == spp and all(x < y for x, y in zip(a, a[1:])), (kw, a, b)
        for sc in (2, 8):  # shard fields never matter
            assert sched(lib.rayz_hip_chunk_schedule, **kw, shard_count=sc, shard_index=1, tile_rows=1) == a
    assert sched(lib.rayz_hip_chunk_schedule, width=64, height=36, samples_per_px=40) == [0, 16, 32, 40]
    big = sched(lib.rayz_hip_chunk_schedule, width=1920, height=1080, samples_per_px=1024)
    # the largest chunk is sized for an 8-way deal of the frame (auto_chunk: pixels x spp / 2^24 held to [64, 256]): 64 here
    assert [y - x for x, y in zip(big, big[1:])] == [64] * 15 + [32, 16, 16]
    c2 = sched(lib.rayz_hip_chunk_schedule, width=1920, height=1080, samples_per_px=256)
    assert [y - x for x, y in zip(c2, c2[1:])] == [64, 64, 64, 32, 16, 16]
    k4 = sched(lib.rayz_hip_chunk_schedule, width=3840, height=2160, samples_per_px=4096)
    assert len(k4) - 1 == 20 and k4[-1] - k4[-2] == 16 and k4[1] == 256  # 3840x2160x4096: big enough for 256-sample chunks
    p = capi.RenderParams(width=4, height=4, samples_per_px=1 << 30, chunk_spp=1)
    t = tracer.threeSpheres(32, seed=1)
    out = np.zeros((4, 4, 3), dtype=np.float32)
    rc = lib.rayz_hip_render(C.byref(t.scene_desc()), C.byref(t.camera_desc()), C.byref(p), out.ctypes.data_as(C.c_void_p), None)
    assert rc == capi.ERR_BAD_ARG and b"chunks per pixel" in lib.rayz_hip_last_error()
    # the automatic schedule (chunk_spp = 0) is bounded the same way: a small frame uses uniform chunks of 16
    p = capi.RenderParams(width=4, height=4, samples_per_px=1 << 25, chunk_spp=0)
    rc = lib.rayz_hip_render(C.byref(t.scene_desc()), C.byref(t.camera_desc()), C.byref(p), out.ctypes.data_as(C.c_void_p), None)
    assert rc == capi.ERR_BAD_ARG and b"chunks per pixel" in lib.rayz_hip_last_error()
    assert lib.rayz_hip_chunk_schedule(C.byref(p), None, 0) == 0  # .. and the schedule query says "none" instead of building 2^21 entries
    p = capi.RenderParams(width=1920, height=1080, samples_per_px=1 << 25, chunk_spp=0)  # 256-sample chunks: 2^17 of them, fine
    assert lib.rayz_hip_chunk_schedule(C.byref(p), None, 0) == (1 << 17) + 4  # 2^17 - 1 chunks of 256, then 128, 64, 32, 16, 16


def test_chunk_count_is_exact_for_every_spp(built, oracle):
    """validate_params bounds the chunks per pixel with chunk_count(), which must never undercount the schedule that
    chunk_schedule() then builds (round 3's `spp / 256 + 8` did, by one, for tails of 256 .. 511 samples: spp = 497 has 10
    chunks).  rayz_hip_chunk_schedule returns 0 when the two disagree; the oracle's independent restatement gives the length."""
    lib, olib = capi.load(), oracle.load()
    buf, obuf = (C.c_uint32 * 640)(), (C.c_uint32 * 640)()
    for spp in list(range(64, 20001)) + [(1 << k) + d for k in range(15, 26) for d in (-1, 0, 1, 255, 256, 257, 497)]:
        p = capi.RenderParams(width=1920, height=1080, samples_per_px=spp)
        n = lib.rayz_hip_chunk_schedule(C.byref(p), buf, 640)
        m = olib.rayz_oracle_chunk_schedule(C.byref(p), obuf, 640)
        assert n == m and n != 0, (spp, n, m)
        k = min(n + 1, 640)
        assert list(buf[:k]) == list(obuf[:k]) and buf[0] == 0 and (n >= 640 or buf[n] == spp), spp
    for spp in range(4000, 9001):  # a frame big enough for 256-sample chunks
        p = capi.RenderParams(width=3840, height=2160, samples_per_px=spp)
        n = lib.rayz_hip_chunk_schedule(C.byref(p), buf, 640)
        assert n == olib.rayz_oracle_chunk_schedule(C.byref(p), obuf, 640) and n != 0 and list(buf[: n + 1]) == list(obuf[: n + 1]), spp
    p = capi.RenderParams(width=3840, height=2160, samples_per_px=30 * 256 + 497)
    assert lib.rayz_hip_chunk_schedule(C.byref(p), buf, 640) == 30 + 10 == (30 * 256 + 497) // 256 + 9  # the case `spp / 256 + 8` undercounted


def test_debug_knobs_that_could_hang_a_kernel_are_refused(built):
    """rayz_hip_debug_set changes scheduling only, and must not be able to hang the device: a threshold byte of 0 lanes
    (round 3: keep_stepping = 0 made trace_kernel_bvh's box-step loop spin with no lane stepping), thresholds above a wave's 64
    lanes, a zero queue reservation and the retired two-path kernel (not in this build) are BAD_ARG; negative = default."""
    lib = capi.load()
    ok = lambda knob, v: lib.rayz_hip_debug_set(knob, v)  # noqa: E731
    try:
        for bad in (20, 20 | (0 << 8), 0 | (18 << 8), 0, 65 | (18 << 8), 20 | (65 << 8), 20 | (18 << 8) | (1 << 16)):
            assert ok(capi.DEBUG_BVH_KEEP, bad) == capi.ERR_BAD_ARG, hex(bad)
            assert b"BVH_KEEP" in lib.rayz_hip_last_error()
        for good in (20 | (18 << 8), 1 | (1 << 8), 64 | (64 << 8), -1):
            assert ok(capi.DEBUG_BVH_KEEP, good) == capi.OK, hex(good)
        assert ok(capi.DEBUG_QUEUE_GRAB, 0) == capi.ERR_BAD_ARG and ok(capi.DEBUG_QUEUE_GRAB, 1 << 21) == capi.ERR_BAD_ARG
        assert ok(capi.DEBUG_QUEUE_GRAB, 64) == capi.OK
        assert ok(capi.DEBUG_BVH_KERNEL, 2) == capi.ERR_BAD_ARG and b"RAYZ_EXPERIMENTS" in lib.rayz_hip_last_error()
        assert ok(capi.DEBUG_BVH_KERNEL, 1) == capi.OK
        assert ok(capi.DEBUG_BVH2_KEEP, 40 | (10 << 8) | (6 << 16) | (18 << 24)) == capi.ERR_BAD_ARG
        assert ok(capi.DEBUG_LDS_PAD, 1 << 20) == capi.ERR_BAD_ARG
        assert ok(99, 1) == capi.ERR_BAD_ARG
    finally:
        for k in (capi.DEBUG_BVH_KEEP, capi.DEBUG_QUEUE_GRAB, capi.DEBUG_BVH_KERNEL, capi.DEBUG_BVH2_KEEP, capi.DEBUG_LDS_PAD):
            lib.rayz_hip_debug_set(k, -1)


def test_automatic_schedule_invariants_over_random_frames(built, oracle):
    """Properties of the automatic chunk schedule over random frame sizes: it partitions [0, spp); chunk sizes never
    grow along the schedule (big items first, the queue ends in short ones); the largest chunk is a power of two, at most 256, at most
    spp / 2, and — round 4 — at most 1/8 of a lane's share of an 8-way deal over 2^18-lane GPUs (pixels x spp / 2^24) unless that would
    go below 64; small frames and low sample counts use uniform 16; library and oracle agree; the shard fields never matter."""
    lib, olib = capi.load(), oracle.load()
    rng = np.random.default_rng(11)
    buf, obuf = (C.c_uint32 * 4096)(), (C.c_uint32 * 4096)()
    for _ in range(600):
        w, h = int(rng.integers(1, 6000)), int(rng.integers(1, 4000))
        spp = int(2 ** rng.uniform(0, 13.5))
        p = capi.RenderParams(width=w, height=h, samples_per_px=spp, shard_index=int(rng.integers(0, 3)), shard_count=3, tile_rows=int(rng.integers(0, 9)))
        n = lib.rayz_hip_chunk_schedule(C.byref(p), buf, 4096)
        assert n == olib.rayz_oracle_chunk_schedule(C.byref(p), obuf, 4096) and 0 < n < 4096, (w, h, spp, n)
        a = list(buf[: n + 1])
        assert a == list(obuf[: n + 1]) and a[0] == 0 and a[-1] == spp
        sizes = [y - x for x, y in zip(a, a[1:])]
        assert all(s > 0 for s in sizes)
        if w * h < (1 << 19) or spp < 64:
            assert all(s == 16 for s in sizes[:-1]) and sizes[-1] <= 16
            continue
        assert all(x >= y for x, y in zip(sizes, sizes[1:])), (w, h, spp, sizes)
        big = sizes[0]
        assert big & (big - 1) == 0 and big <= 256 and big <= spp // 2
        assert big <= max(64, (w * h * spp) >> 24), (w, h, spp, big)
        assert min(sizes[:-1]) >= 16  # (only the last chunk may be a remainder)
