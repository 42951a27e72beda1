"""ctypes loader of the CPU oracle (oracle/librayz_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by rayz_amd/.  It borrows the ABI structure definitions from rayz_amd.capi because the oracle
consumes the very PODs the C ABI defines (include/rayz_hip.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rayz_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librayz_oracle.so")

_P = C.POINTER
_D = _P(C.c_double)
_U64 = _P(C.c_uint64)
_lib = None


def build(force: bool = False) -> str:
    # make decides (the library depends on rayz_oracle.cpp AND include/rayz_hip.h)
    subprocess.run(["make", "-C", HERE, "-s"] + (["-B"] if force else []), check=True)
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    S, Cm, Pm, St = _P(capi.SceneDesc), _P(capi.CameraDesc), _P(capi.RenderParams), _P(capi.RenderStats)
    sig = {
        "rayz_oracle_render_b_f32": (C.c_int, [S, Cm, Pm, _P(C.c_uint32), C.c_uint32, C.c_void_p, St, C.c_int]),
        "rayz_oracle_render_b_f64": (C.c_int, [S, Cm, Pm, _P(C.c_uint32), C.c_uint32, C.c_void_p, St, C.c_int]),
        "rayz_oracle_shard_rows": (C.c_uint32, [Pm]),
        "rayz_oracle_chunk_schedule": (C.c_uint32, [Pm, _P(C.c_uint32), C.c_uint32]),
        "rayz_oracle_kat_b": (C.c_int, [C.c_uint32, C.c_uint32, _D, C.c_uint32, _D]),
        "rayz_oracle_kat_a": (C.c_int, [C.c_uint32, _D, C.c_uint32, _D]),
        "rayz_oracle_filter_audit": (C.c_int, [S, Cm, Pm, C.c_uint32, _P(C.c_uint32), C.c_uint32, _U64]),
        "rayz_oracle_render_a": (C.c_int, [S, Cm, Pm, C.c_uint32, C.c_uint32, _U64, C.c_int, _D, _D, St]),
        "rayz_oracle_bvh_flat": (C.c_uint32, [S, _D, _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32)]),
        "rayz_oracle_camera_init": (None, [C.c_double, C.c_double, C.c_double, _D, _D, _D, C.c_uint32, C.c_uint32, Cm]),
        "rayz_oracle_get_ray_norng": (None, [Cm, C.c_uint32, C.c_uint32, _D, _D]),
        "rayz_oracle_refract": (None, [_D, _D, C.c_double, _D]),
        "rayz_oracle_reflectance": (C.c_double, [C.c_double, C.c_double]),
        "rayz_oracle_sphere_bbox": (None, [_P(capi.Sphere), _D, _D]),
        "rayz_oracle_aabb_enclose": (None, [_D, _D, _D, _D, _D, _D]),
        "rayz_oracle_aabb_hit": (C.c_int, [_D, _D, _D, _D, C.c_double, C.c_double]),
        "rayz_oracle_v3_op": (None, [C.c_int, _D, _D, _D]),
        "rayz_oracle_v3_dot": (C.c_double, [_D, _D]),
        "rayz_oracle_v3_mag": (C.c_double, [_D]),
        "rayz_oracle_v3_amax": (C.c_int, [_D]),
        "rayz_oracle_ppm_u8": (None, [_D, _P(C.c_uint8)]),
        "rayz_oracle_write_ppm": (C.c_int, [C.c_char_p, _D, C.c_uint32, C.c_uint32]),
        "rayz_oracle_splitmix64": (None, [C.c_uint64, C.c_uint32, _U64]),
        "rayz_oracle_xoshiro_seed": (None, [C.c_uint64, _U64]),
        "rayz_oracle_xoshiro_u64": (None, [_U64, C.c_uint32, _U64]),
        "rayz_oracle_xoshiro_f64": (None, [_U64, C.c_uint32, _D]),
        "rayz_oracle_pcg32": (None, [C.c_uint64, C.c_uint64, C.c_uint32, _P(C.c_uint32)]),
        "rayz_oracle_pcg32_path": (None, [C.c_uint64, C.c_uint64, C.c_uint32, _P(C.c_uint32)]),
        "rayz_oracle_random_bouncing": (C.c_int, [_U64, C.c_int, C.c_int, _P(capi.Sphere), C.c_uint32,
                                                  _P(capi.Material), C.c_uint32, _P(capi.Texture), C.c_uint32,
                                                  _P(C.c_uint32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def d3(v) -> C.Array:
    return capi.D3(*[float(x) for x in v])


def kat_b(op: int, records: np.ndarray, precision: int = capi.PRECISION_F64) -> np.ndarray:
    """Mode B's pieces on rayz_hip_kat records: (n, KAT_IN_STRIDE) float64 -> (n, KAT_OUT_STRIDE)."""
    rec = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, capi.KAT_IN_STRIDE)
    out = np.zeros((len(rec), capi.KAT_OUT_STRIDE))
    rc = load().rayz_oracle_kat_b(op, precision, rec.ctypes.data_as(_D), len(rec), out.ctypes.data_as(_D))
    assert rc == 0, rc
    return out


def kat_a(op: int, records: np.ndarray) -> np.ndarray:
    """The same records through mode A (the reference's own functions, f64)."""
    rec = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, capi.KAT_IN_STRIDE)
    out = np.zeros((len(rec), capi.KAT_OUT_STRIDE))
    rc = load().rayz_oracle_kat_a(op, rec.ctypes.data_as(_D), len(rec), out.ctypes.data_as(_D))
    assert rc == 0, rc
    return out


def filter_audit(scene, camera, params, pixels, precision: int = capi.PRECISION_F32) -> dict:
    pl = np.ascontiguousarray(pixels, dtype=np.uint32)
    out = np.zeros(5, dtype=np.uint64)
    rc = load().rayz_oracle_filter_audit(C.byref(scene), C.byref(camera), C.byref(params), precision,
                                         pl.ctypes.data_as(_P(C.c_uint32)), len(pl), out.ctypes.data_as(_U64))
    assert rc == 0, rc
    return dict(zip(("pairs", "candidates", "f64_hits", "false_negatives", "unpadded_false_negatives"), map(int, out)))


def render_b(scene, camera, params, pixels=None, threads: int = 0):
    """Mode B (kernel arithmetic).  Returns (image or (n,3) pixel list, stats)."""
    lib = load()
    f64 = params.precision == capi.PRECISION_F64
    dt = np.float64 if f64 else np.float32
    if pixels is None:
        rows = lib.rayz_oracle_shard_rows(C.byref(params))
        out = np.empty((rows, params.width, 3), dtype=dt)
        plist, n = None, 0
    else:
        pl = np.ascontiguousarray(pixels, dtype=np.uint32)
        out = np.empty((len(pl), 3), dtype=dt)
        plist, n = pl.ctypes.data_as(_P(C.c_uint32)), len(pl)
    st = capi.RenderStats()
    fn = lib.rayz_oracle_render_b_f64 if f64 else lib.rayz_oracle_render_b_f32
    rc = fn(C.byref(scene), C.byref(camera), C.byref(params), plist, n, out.ctypes.data_as(C.c_void_p), C.byref(st),
            threads)
    if rc != 0:
        raise RuntimeError(f"oracle mode B failed: {rc}")
    return out, st


def render_a(scene, camera, params, rng_state, row_begin=0, row_end=None, linear=False, want_sumsq=False):
    """Mode A (the reference as written, f64, one sequential stream).  rng_state: 4 u64, updated in place."""
    lib = load()
    row_end = params.height if row_end is None else row_end
    out = np.empty((row_end - row_begin, params.width, 3), dtype=np.float64)
    sq = np.empty_like(out) if want_sumsq else None
    st = capi.RenderStats()
    state = np.ascontiguousarray(rng_state, dtype=np.uint64)
    rc = lib.rayz_oracle_render_a(C.byref(scene), C.byref(camera), C.byref(params), row_begin, row_end,
                                  state.ctypes.data_as(_U64), 1 if linear else 0, out.ctypes.data_as(_D),
                                  sq.ctypes.data_as(_D) if want_sumsq else None, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"oracle mode A failed: {rc}")
    rng_state[:] = state
    return (out, sq, st) if want_sumsq else (out, st)


class OracleScene:
    """A pool generated by the oracle's own `randomBouncing` restatement (to cross-check the product's)."""

    def __init__(self, seed: int, lo: int = -11, hi: int = 11):
        lib = load()
        cap = (hi - lo) * (hi - lo) + 8
        self.spheres = (capi.Sphere * cap)()
        self.materials = (capi.Material * cap)()
        self.textures = (capi.Texture * cap)()
        self.rng_state = np.zeros(4, dtype=np.uint64)
        lib.rayz_oracle_xoshiro_seed(seed, self.rng_state.ctypes.data_as(_U64))
        counts = (C.c_uint32 * 3)()
        rc = lib.rayz_oracle_random_bouncing(self.rng_state.ctypes.data_as(_U64), lo, hi, self.spheres, cap,
                                             self.materials, cap, self.textures, cap, counts)
        if rc != 0:
            raise RuntimeError("oracle randomBouncing failed")
        self.counts = tuple(counts)
        self.desc = capi.SceneDesc(spheres=self.spheres, materials=self.materials, textures=self.textures,
                                   n_spheres=counts[0], n_materials=counts[1], n_textures=counts[2])
