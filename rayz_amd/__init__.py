"""rayz_amd — MI355X (gfx950) implementation of the `Tracer.render()` hot path of jlucier/rayz.

Layout:
  csrc/     hand-written HIP kernels + the C ABI of include/rayz_hip.h (librayz_hip.so)
  host/     C++ mirror of the reference's Tracer / MemPool / Camera / Image API + the `rayz` CLI
  zig/      the Zig-side drop-in for src/renderer.zig (source only: no zig toolchain in this image)
  capi.py   ctypes view of the C ABI          tracer.py  Python view of the host mirror
  render.py device-resident rendering into caller-owned GPU buffers (bench / multi-GPU shards)

There is no CPU fallback: every render goes through the HIP library, and loading fails loudly when
it has not been built.
"""
from . import capi  # noqa: F401
from .capi import RayzHipError  # noqa: F401

__all__ = ["capi", "RayzHipError"]
