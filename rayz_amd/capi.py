"""ctypes view of include/rayz_hip.h — the C ABI of the MI355X render path.

The structures mirror the header field for field; `load()` opens the in-tree
`librayz_hip.so` built by `__graft_entry__.build()` and fails loudly when it is
missing: there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "librayz_hip.so")

ABI_VERSION = 5
MAX_DEVICES = 64
GATHER_RCCL, GATHER_PEER_COPY = 0, 1
GATHER_ALLOW_DUPLICATE_DEVICES = 0x100  # flag bit, peer-copy only (tests on a one-GPU box)
(DEBUG_QUEUE_GRAB, DEBUG_BVH_KEEP, DEBUG_BVH_PEEL, DEBUG_BVH_TOP, DEBUG_BVH_KERNEL, DEBUG_BVH2_KEEP,
 DEBUG_LDS_PAD, DEBUG_BVH_TOP_ORDER, DEBUG_BVH_NODES, DEBUG_BVH_SPLIT, DEBUG_BVHX, DEBUG_CHUNK_CAP) = range(12)
(KAT_REFRACT, KAT_REFLECTANCE, KAT_GET_RAY, KAT_BOX_HIT, KAT_SPHERE_HIT, KAT_SCATTER, KAT_CHECKER, KAT_BACKGROUND,
 KAT_TRIANGLE_HIT, KAT_SCAN_DISCS) = range(10)
KAT_IN_STRIDE, KAT_OUT_STRIDE = 48, 12

OK = 0
ERR_BAD_ARG = -1
ERR_HIP = -2
ERR_OOM = -3
ERR_NO_DEVICE = -4
ERR_STATE = -5

TEX_CHECKER, TEX_SOLID = 0, 1
MAT_DIFFUSE, MAT_METALLIC, MAT_DIELECTRIC = 0, 1, 2
DIFFUSE_UNIT_SPHERE, DIFFUSE_UNIT_SPHERE_SURFACE, DIFFUSE_HEMISPHERE = 0, 1, 2
PRECISION_F32, PRECISION_F64 = 0, 1
TRAVERSAL_LINEAR, TRAVERSAL_BVH, TRAVERSAL_AUTO = 0, 1, 2

D3 = C.c_double * 3


class Texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("even", C.c_uint32), ("odd", C.c_uint32), ("_pad", C.c_uint32),
                ("scale", C.c_double), ("color", D3)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("method", C.c_uint32), ("_pad", C.c_uint32),
                ("param", C.c_double)]


class Sphere(C.Structure):
    _fields_ = [("center", D3), ("velocity", D3), ("radius", C.c_double), ("material", C.c_uint32),
                ("_pad", C.c_uint32)]


class Triangle(C.Structure):
    _fields_ = [("v0", D3), ("v1", D3), ("v2", D3), ("material", C.c_uint32), ("_pad", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("spheres", C.POINTER(Sphere)), ("materials", C.POINTER(Material)),
                ("textures", C.POINTER(Texture)), ("n_spheres", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_textures", C.c_uint32), ("n_triangles", C.c_uint32), ("triangles", C.POINTER(Triangle))]


class CameraDesc(C.Structure):
    _fields_ = [("look_from", D3), ("px_du", D3), ("px_dv", D3), ("px_origin", D3), ("defocus_u", D3),
                ("defocus_v", D3), ("defocus", C.c_uint32), ("_pad", C.c_uint32)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_px", C.c_uint32),
                ("max_bounces", C.c_uint32), ("seed", C.c_uint64), ("tmin", C.c_double),
                ("precision", C.c_uint32), ("traversal", C.c_uint32), ("chunk_spp", C.c_uint32),
                ("tile_rows", C.c_uint32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32)]


class RenderStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("segments", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("node_tests", C.c_uint64), ("kernel_ms", C.c_double)]


assert C.sizeof(Texture) == 48 and C.sizeof(Material) == 24 and C.sizeof(Sphere) == 64 and C.sizeof(Triangle) == 80
assert C.sizeof(SceneDesc) == 48
assert C.sizeof(CameraDesc) == 152 and C.sizeof(RenderParams) == 56 and C.sizeof(RenderStats) == 40

# every symbol include/rayz_hip.h declares: (name, restype, argtypes)
PROTOTYPES = [
    ("rayz_hip_init", C.c_int, [C.c_int]),
    ("rayz_hip_shutdown", None, []),
    ("rayz_hip_last_error", C.c_char_p, []),
    ("rayz_hip_abi_version", C.c_uint32, []),
    ("rayz_hip_debug_set", C.c_int, [C.c_uint32, C.c_longlong]),
    ("rayz_hip_shard_rows", C.c_uint32, [C.POINTER(RenderParams)]),
    ("rayz_hip_chunk_schedule", C.c_uint32, [C.POINTER(RenderParams), C.POINTER(C.c_uint32), C.c_uint32]),
    ("rayz_hip_scene_create", C.c_int, [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]),
    ("rayz_hip_scene_create_on", C.c_int, [C.c_int, C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]),
    ("rayz_hip_scene_destroy", C.c_int, [C.c_void_p]),
    ("rayz_hip_scene_bvh", C.c_int,
     [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint32),
      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("rayz_hip_render_device", C.c_int,
     [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.c_void_p]),
    ("rayz_hip_render_device_f64", C.c_int,
     [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.c_void_p]),
    ("rayz_hip_scene_sync", C.c_int, [C.c_void_p, C.POINTER(RenderStats)]),
    ("rayz_hip_render", C.c_int,
     [C.POINTER(SceneDesc), C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p,
      C.POINTER(RenderStats)]),
    ("rayz_hip_render_f64", C.c_int,
     [C.POINTER(SceneDesc), C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p,
      C.POINTER(RenderStats)]),
    ("rayz_hip_tonemap_u8", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rayz_hip_kat", C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_double)]),
    ("rayz_hip_multi_create", C.c_int,
     [C.POINTER(C.c_int), C.c_int, C.POINTER(SceneDesc), C.c_uint32, C.POINTER(C.c_void_p)]),
    ("rayz_hip_multi_destroy", C.c_int, [C.c_void_p]),
    ("rayz_hip_multi_info", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    ("rayz_hip_multi_device_stats", C.c_int, [C.c_void_p, C.c_int, C.POINTER(RenderStats)]),
    ("rayz_hip_multi_timing", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("rayz_hip_multi_render", C.c_int,
     [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.POINTER(RenderStats)]),
    ("rayz_hip_multi_render_f64", C.c_int,
     [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.POINTER(RenderStats)]),
    ("rayz_hip_multi_render_u8", C.c_int,
     [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.POINTER(RenderStats)]),
    ("rayz_hip_render_multi", C.c_int,
     [C.POINTER(C.c_int), C.c_int, C.POINTER(SceneDesc), C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p,
      C.POINTER(RenderStats)]),
    ("rayz_hip_render_multi_f64", C.c_int,
     [C.POINTER(C.c_int), C.c_int, C.POINTER(SceneDesc), C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p,
      C.POINTER(RenderStats)]),
]


class RayzHipError(RuntimeError):
    pass


_lib = None


def load(path: str | None = None) -> C.CDLL:
    """Open librayz_hip.so and bind every prototype.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RayzHipError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "rayz_amd has no CPU fallback.")
    lib = C.CDLL(p)
    for name, res, args in PROTOTYPES:
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.rayz_hip_abi_version() != ABI_VERSION:
        raise RayzHipError(f"ABI mismatch: library {lib.rayz_hip_abi_version()} != binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(lib: C.CDLL, rc: int, what: str) -> None:
    if rc != OK:
        msg = lib.rayz_hip_last_error()
        raise RayzHipError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")
