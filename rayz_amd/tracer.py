"""Python view of the host mirror (include/rayz_host.h): rayz's Tracer / MemPool / Image API.

Thin ctypes wrappers over the C++ mirror in rayz_amd/host/rayz.hpp, with the reference's names
(`Tracer.init`, `tracer.pool.add...`, `tracer.render()`, `tracer.img.writePPM`), so tests and
bench.py read like a caller of jlucier/rayz (src/rayz.zig:12-168).  The render goes through
include/rayz_hip.h on the GPU; nothing here computes an image.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

FIELD_MAX_BOUNCES, FIELD_SAMPLES_PER_PX, FIELD_PRECISION, FIELD_TRAVERSAL = 0, 1, 2, 3
FIELD_CHUNK_SPP, FIELD_RENDER_SEED, FIELD_TMIN = 4, 5, 6


class TracerInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_px", C.c_uint32),
                ("max_bounces", C.c_uint32), ("n_spheres", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_textures", C.c_uint32), ("n_triangles", C.c_uint32)]


_P = C.POINTER
_D = _P(C.c_double)
HOST_PROTOTYPES = [
    ("rayz_tracer_create", C.c_int,
     [C.c_uint32, C.c_double, C.c_double, C.c_double, _D, _D, _D, C.c_int, C.c_uint64, _P(C.c_void_p)]),
    ("rayz_tracer_destroy", None, [C.c_void_p]),
    ("rayz_tracer_add_texture_solid", C.c_int64, [C.c_void_p, _D]),
    ("rayz_tracer_add_texture_checker", C.c_int64, [C.c_void_p, C.c_double, C.c_uint32, C.c_uint32]),
    ("rayz_tracer_add_material_diffuse", C.c_int64, [C.c_void_p, C.c_uint32, C.c_uint32]),
    ("rayz_tracer_add_material_metallic", C.c_int64, [C.c_void_p, C.c_uint32, C.c_double]),
    ("rayz_tracer_add_material_dielectric", C.c_int64, [C.c_void_p, C.c_double]),
    ("rayz_tracer_add_sphere", C.c_int64, [C.c_void_p, _D, _D, C.c_double, C.c_uint32]),
    ("rayz_tracer_add_triangle", C.c_int64, [C.c_void_p, _D, _D, _D, C.c_uint32]),
    ("rayz_tracer_set_u64", C.c_int, [C.c_void_p, C.c_int, C.c_uint64]),
    ("rayz_tracer_set_f64", C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    ("rayz_tracer_set_devices", C.c_int, [C.c_void_p, _P(C.c_int), C.c_int]),
    ("rayz_tracer_info", C.c_int, [C.c_void_p, _P(TracerInfo)]),
    ("rayz_tracer_camera", C.c_int, [C.c_void_p, _P(capi.CameraDesc)]),
    ("rayz_tracer_get_ray", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, _D, _D]),
    ("rayz_tracer_scene", C.c_int, [C.c_void_p, _P(capi.SceneDesc)]),
    ("rayz_tracer_params", C.c_int, [C.c_void_p, _P(capi.RenderParams)]),
    ("rayz_tracer_rng_state", C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    ("rayz_tracer_rng_next", C.c_uint64, [C.c_void_p]),
    ("rayz_tracer_rng_float", C.c_double, [C.c_void_p]),
    ("rayz_tracer_render", C.c_int64, [C.c_void_p]),
    ("rayz_tracer_stats", C.c_int, [C.c_void_p, _P(capi.RenderStats)]),
    ("rayz_tracer_pixels", _D, [C.c_void_p]),
    ("rayz_tracer_write_ppm", C.c_int, [C.c_void_p, C.c_char_p]),
    ("rayz_image_write_ppm", C.c_int, [_D, C.c_uint32, C.c_uint32, C.c_char_p]),
    ("rayz_image_to_u8", None, [_D, C.c_size_t, _P(C.c_uint8)]),
    ("rayz_scene_random_bouncing", C.c_int, [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint64, _P(C.c_void_p)]),
    ("rayz_scene_three_spheres", C.c_int, [C.c_uint32, C.c_int, C.c_uint64, _P(C.c_void_p)]),
    ("rayz_scene_triangle_mesh", C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, _P(C.c_void_p)]),
]

_bound = False


def _lib() -> C.CDLL:
    global _bound
    lib = capi.load()
    if not _bound:
        for name, res, args in HOST_PROTOTYPES:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return lib


def _d3(v) -> C.Array:
    return capi.D3(*[float(x) for x in v])


class MemPool:
    """`MemPool`, src/ecs.zig:22-69: handles are plain indices."""

    def __init__(self, tracer: "Tracer"):
        self._t = tracer

    def _h(self, v: int, what: str) -> int:
        if v < 0:
            raise capi.RayzHipError(f"pool.add {what} failed (status {v})")
        return int(v)

    def add_solid_texture(self, color) -> int:
        return self._h(_lib().rayz_tracer_add_texture_solid(self._t._h, _d3(color)), "texture")

    def add_checker_texture(self, scale: float, even: int, odd: int) -> int:
        return self._h(_lib().rayz_tracer_add_texture_checker(self._t._h, scale, even, odd), "texture")

    def add_diffuse(self, texture: int, method: int = capi.DIFFUSE_HEMISPHERE) -> int:
        return self._h(_lib().rayz_tracer_add_material_diffuse(self._t._h, texture, method), "material")

    def add_metallic(self, texture: int, fuzz: float = 0.0) -> int:
        return self._h(_lib().rayz_tracer_add_material_metallic(self._t._h, texture, fuzz), "material")

    def add_dielectric(self, refractive_index: float = 1.0) -> int:
        return self._h(_lib().rayz_tracer_add_material_dielectric(self._t._h, refractive_index), "material")

    def add_sphere(self, center, radius: float, material: int, velocity=(0.0, 0.0, 0.0)) -> int:
        return self._h(_lib().rayz_tracer_add_sphere(self._t._h, _d3(center), _d3(velocity), radius, material),
                       "sphere")


    def add_triangle(self, v0, v1, v2, material: int) -> int:
        """Build-defined triangle hittable (the reference has spheres only)."""
        return self._h(_lib().rayz_tracer_add_triangle(self._t._h, _d3(v0), _d3(v1), _d3(v2), material), "triangle")


class Image:
    """`Image`, src/image.zig:4-41: `pixels` is (h, w, 3) float64 linear radiance."""

    def __init__(self, h: int, w: int):
        self.h, self.w = h, w
        self.pixels = np.zeros((h, w, 3), dtype=np.float64)

    def writePPM(self, path: str) -> None:
        px = np.ascontiguousarray(self.pixels, dtype=np.float64)
        rc = _lib().rayz_image_write_ppm(px.ctypes.data_as(_D), self.w, self.h, path.encode())
        if rc != capi.OK:
            raise capi.RayzHipError(f"writePPM({path}) failed (status {rc})")

    def to_u8(self) -> np.ndarray:
        px = np.ascontiguousarray(self.pixels, dtype=np.float64)
        out = np.empty((self.h, self.w, 3), dtype=np.uint8)
        _lib().rayz_image_to_u8(px.ctypes.data_as(_D), self.h * self.w, out.ctypes.data_as(_P(C.c_uint8)))
        return out


class Tracer:
    """`Tracer`, src/renderer.zig:18-101."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)
        self.pool = MemPool(self)
        info = self.info()
        self.img = Image(info.height, info.width)
        self.stats = capi.RenderStats()

    @classmethod
    def init(cls, img_w: int, vfov: float, focus_dist: float, defocus_angle: float, look_from, look_at, vup,
             seed: int | None = None) -> "Tracer":
        h = C.c_void_p()
        rc = _lib().rayz_tracer_create(img_w, vfov, focus_dist, defocus_angle, _d3(look_from), _d3(look_at),
                                       _d3(vup), 0 if seed is None else 1, seed or 0, C.byref(h))
        if rc != capi.OK:
            raise capi.RayzHipError(f"Tracer.init failed (status {rc})")
        return cls(h.value)

    def __del__(self):
        try:
            if self._h:
                _lib().rayz_tracer_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- fields --
    def info(self) -> TracerInfo:
        i = TracerInfo()
        _lib().rayz_tracer_info(self._h, C.byref(i))
        return i

    def _set(self, field: int, v: int) -> None:
        rc = _lib().rayz_tracer_set_u64(self._h, field, int(v))
        if rc != capi.OK:
            raise capi.RayzHipError(f"bad value {v} for tracer field {field}")

    samples_per_px = property(lambda s: s.info().samples_per_px, lambda s, v: s._set(FIELD_SAMPLES_PER_PX, v))
    max_bounces = property(lambda s: s.info().max_bounces, lambda s, v: s._set(FIELD_MAX_BOUNCES, v))

    def set_gpu(self, precision: int | None = None, traversal: int | None = None, chunk_spp: int | None = None,
                render_seed: int | None = None, tmin: float | None = None, devices=None) -> "Tracer":
        if precision is not None:
            self._set(FIELD_PRECISION, precision)
        if traversal is not None:
            self._set(FIELD_TRAVERSAL, traversal)
        if chunk_spp is not None:
            self._set(FIELD_CHUNK_SPP, chunk_spp)
        if render_seed is not None:
            self._set(FIELD_RENDER_SEED, render_seed)
        if tmin is not None:
            _lib().rayz_tracer_set_f64(self._h, FIELD_TMIN, tmin)
        if devices is not None:  # render() then drives all of them through rayz_hip_render_multi
            arr = (C.c_int * len(devices))(*devices)
            if _lib().rayz_tracer_set_devices(self._h, arr, len(devices)) != capi.OK:
                raise capi.RayzHipError(f"bad device list {devices}")
        return self

    # -- what render() hands to the C ABI --
    def camera_desc(self) -> capi.CameraDesc:
        c = capi.CameraDesc()
        _lib().rayz_tracer_camera(self._h, C.byref(c))
        return c

    def scene_desc(self) -> capi.SceneDesc:
        """Borrowed view of the flattened pool; valid until the tracer is mutated or dropped."""
        s = capi.SceneDesc()
        rc = _lib().rayz_tracer_scene(self._h, C.byref(s))
        if rc != capi.OK:
            raise capi.RayzHipError(f"flatten failed (status {rc})")
        s._owner = self  # keep the tracer alive while the view is
        return s

    def params(self) -> capi.RenderParams:
        p = capi.RenderParams()
        _lib().rayz_tracer_params(self._h, C.byref(p))
        return p

    def get_ray(self, px: int, py: int):
        o, d = capi.D3(), capi.D3()
        _lib().rayz_tracer_get_ray(self._h, px, py, o, d)
        return np.array(o), np.array(d)

    def rng_state(self) -> np.ndarray:
        s = (C.c_uint64 * 4)()
        _lib().rayz_tracer_rng_state(self._h, s)
        return np.array(s, dtype=np.uint64)

    def rng_next(self) -> int:
        return int(_lib().rayz_tracer_rng_next(self._h))

    def rng_float(self) -> float:
        return float(_lib().rayz_tracer_rng_float(self._h))

    # -- the path --
    def render(self) -> int:
        lib = _lib()
        rays = lib.rayz_tracer_render(self._h)
        if rays < 0:
            capi.check(lib, int(rays), "Tracer.render")
        lib.rayz_tracer_stats(self._h, C.byref(self.stats))
        p = lib.rayz_tracer_pixels(self._h)
        n = self.img.h * self.img.w * 3
        self.img.pixels = np.ctypeslib.as_array(p, shape=(n,)).reshape(self.img.h, self.img.w, 3).copy()
        return int(rays)


def randomBouncing(img_w: int, grid_lo: int = -11, grid_hi: int = 11, seed: int | None = None) -> Tracer:
    """`randomBouncing`, src/rayz.zig:45-168 (grid bounds as parameters)."""
    h = C.c_void_p()
    rc = _lib().rayz_scene_random_bouncing(img_w, grid_lo, grid_hi, 0 if seed is None else 1, seed or 0, C.byref(h))
    if rc != capi.OK:
        raise capi.RayzHipError(f"randomBouncing failed (status {rc})")
    return Tracer(h.value)


def triangleMesh(img_w: int, n: int = 224, seed: int | None = None) -> Tracer:
    """BASELINE config 5 (build-defined): n x n-quad height field (2 n^2 triangles) + three spheres."""
    h = C.c_void_p()
    rc = _lib().rayz_scene_triangle_mesh(img_w, n, 0 if seed is None else 1, seed or 0, C.byref(h))
    if rc != capi.OK:
        raise capi.RayzHipError(f"triangleMesh failed (status {rc})")
    return Tracer(h.value)


def threeSpheres(img_w: int, seed: int | None = None) -> Tracer:
    """BASELINE config 1: three stationary Lambertian spheres."""
    h = C.c_void_p()
    rc = _lib().rayz_scene_three_spheres(img_w, 0 if seed is None else 1, seed or 0, C.byref(h))
    if rc != capi.OK:
        raise capi.RayzHipError(f"threeSpheres failed (status {rc})")
    return Tracer(h.value)
