"""Device-resident rendering through the C ABI (include/rayz_hip.h).

`DeviceScene` keeps the flattened pool in HBM and renders shards into GPU buffers the caller owns
(torch tensors in bench.py and the tests) — the split form of `Tracer.render()` that keeps uploads,
allocation and the host copy out of the timed region.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def init(device: int = 0) -> None:
    lib = capi.load()
    capi.check(lib, lib.rayz_hip_init(device), f"rayz_hip_init({device})")


def debug_set(knob: int, value: int = -1) -> None:
    """`rayz_hip_debug_set`: a measurement knob (scheduling / walked tree; never an image); value < 0 = default."""
    lib = capi.load()
    capi.check(lib, lib.rayz_hip_debug_set(knob, value), f"rayz_hip_debug_set({knob})")


def shard_rows(params: capi.RenderParams) -> int:
    return int(capi.load().rayz_hip_shard_rows(C.byref(params)))


def shard_row_indices(height: int, tile_rows: int, shard_index: int, shard_count: int) -> np.ndarray:
    """Global row numbers of a shard's compact output, in order (host-side bookkeeping of the gather)."""
    tile_rows = tile_rows or 8  # RAYZ_DEFAULT_TILE_ROWS
    shard_count = shard_count or 1
    rows = np.arange(height)
    return rows[(rows // tile_rows) % shard_count == shard_index]


def render_host(scene: capi.SceneDesc, camera: capi.CameraDesc, params: capi.RenderParams):
    """One-shot blocking render into host memory: `rayz_hip_render[_f64]`.  Returns (image, stats)."""
    lib = capi.load()
    rows = shard_rows(params)
    f64 = params.precision == capi.PRECISION_F64
    out = np.empty((rows, params.width, 3), dtype=np.float64 if f64 else np.float32)
    st = capi.RenderStats()
    fn = lib.rayz_hip_render_f64 if f64 else lib.rayz_hip_render
    rc = fn(C.byref(scene), C.byref(camera), C.byref(params), out.ctypes.data_as(C.c_void_p), C.byref(st))
    capi.check(lib, rc, "rayz_hip_render")
    return out, st


class DeviceScene:
    """A pool resident in HBM (`rayz_hip_scene_create`; `device` binds it to that HIP ordinal now)."""

    def __init__(self, scene: capi.SceneDesc, device: int | None = None):
        self._lib = capi.load()
        self._h = C.c_void_p()
        if device is None:
            rc = self._lib.rayz_hip_scene_create(C.byref(scene), C.byref(self._h))
        else:
            rc = self._lib.rayz_hip_scene_create_on(device, C.byref(scene), C.byref(self._h))
        capi.check(self._lib, rc, "rayz_hip_scene_create")

    def render_into(self, camera: capi.CameraDesc, params: capi.RenderParams, out_ptr: int, stream: int = 0) -> None:
        """Asynchronous on `stream` (a hipStream_t as int, 0 = the library's stream); `out_ptr` is device memory."""
        fn = (self._lib.rayz_hip_render_device_f64 if params.precision == capi.PRECISION_F64
              else self._lib.rayz_hip_render_device)
        rc = fn(self._h, C.byref(camera), C.byref(params), C.c_void_p(out_ptr), C.c_void_p(stream))
        capi.check(self._lib, rc, "rayz_hip_render_device")

    def sync(self) -> capi.RenderStats:
        st = capi.RenderStats()
        capi.check(self._lib, self._lib.rayz_hip_scene_sync(self._h, C.byref(st)), "rayz_hip_scene_sync")
        return st

    def close(self) -> None:
        if self._h:
            self._lib.rayz_hip_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiScene:
    """The pool replicated on several GPUs of the node (`rayz_hip_multi_create`): one call renders the whole frame —
    rows dealt to the devices in interleaved tiles, one RCCL gather (or peer copies) to devices[0], host output."""

    def __init__(self, scene: capi.SceneDesc, devices, transport: int = capi.GATHER_RCCL):
        self._lib = capi.load()
        self._h = C.c_void_p()
        self.devices = list(devices)
        arr = (C.c_int * len(self.devices))(*self.devices)
        capi.check(self._lib, self._lib.rayz_hip_multi_create(arr, len(self.devices), C.byref(scene), transport,
                                                              C.byref(self._h)), "rayz_hip_multi_create")

    def info(self):
        n, tr, ver = C.c_int(), C.c_uint32(), C.c_int()
        capi.check(self._lib, self._lib.rayz_hip_multi_info(self._h, C.byref(n), C.byref(tr), C.byref(ver)),
                   "rayz_hip_multi_info")
        return {"n_devices": n.value, "transport": tr.value, "rccl_version": ver.value}

    def render(self, camera: capi.CameraDesc, params: capi.RenderParams, u8: bool = False):
        """Returns (frame (h, w, 3), stats); `u8` gives writePPM's bytes (tone-mapped on each device before the gather)."""
        f64 = params.precision == capi.PRECISION_F64
        out = np.empty((params.height, params.width, 3), dtype=np.uint8 if u8 else (np.float64 if f64 else np.float32))
        st = capi.RenderStats()
        fn = (self._lib.rayz_hip_multi_render_u8 if u8 else
              self._lib.rayz_hip_multi_render_f64 if f64 else self._lib.rayz_hip_multi_render)
        rc = fn(self._h, C.byref(camera), C.byref(params), out.ctypes.data_as(C.c_void_p), C.byref(st))
        capi.check(self._lib, rc, "rayz_hip_multi_render")
        return out, st

    def device_stats(self):
        """Per-device counters of the last frame (each device's own trace-kernel time: shard imbalance shows here)."""
        out = []
        for i in range(len(self.devices)):
            st = capi.RenderStats()
            capi.check(self._lib, self._lib.rayz_hip_multi_device_stats(self._h, i, C.byref(st)), "rayz_hip_multi_device_stats")
            out.append(st)
        return out

    def timing(self):
        """(gather_ms, frame_ms) of the last frame: see include/rayz_hip.h."""
        g, f = C.c_double(), C.c_double()
        capi.check(self._lib, self._lib.rayz_hip_multi_timing(self._h, C.byref(g), C.byref(f)), "rayz_hip_multi_timing")
        return g.value, f.value

    def close(self) -> None:
        if self._h:
            self._lib.rayz_hip_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kat(op: int, records, precision: int = capi.PRECISION_F32) -> np.ndarray:
    """`rayz_hip_kat`: the trace kernels' own device functions on (n, KAT_IN_STRIDE) float64 records."""
    lib = capi.load()
    rec = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, capi.KAT_IN_STRIDE)
    out = np.zeros((len(rec), capi.KAT_OUT_STRIDE))
    D = C.POINTER(C.c_double)
    capi.check(lib, lib.rayz_hip_kat(op, precision, rec.ctypes.data_as(D), len(rec), out.ctypes.data_as(D)), "rayz_hip_kat")
    return out


def tonemap_u8(rgb_ptr: int, out_ptr: int, n_pixels: int, stream: int = 0) -> None:
    lib = capi.load()
    capi.check(lib, lib.rayz_hip_tonemap_u8(C.c_void_p(rgb_ptr), C.c_void_p(out_ptr), n_pixels, C.c_void_p(stream)),
               "rayz_hip_tonemap_u8")
