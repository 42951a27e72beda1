#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz: small seeded inputs (flattened pool, camera, params as raw ABI bytes) with
the images the oracle produces for them.  The reference (Zig) cannot run here and has no settable seed, so
these vectors come from this repository's own CPU restatement (oracle mode B = the kernel's arithmetic,
mode A = the reference as written); they pin the oracle against drift and give the GPU tests fixed targets.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import binding as oracle  # noqa: E402
from rayz_amd import capi, tracer  # noqa: E402


def raw(ptr, T, n):
    return np.frombuffer(C.string_at(ptr, C.sizeof(T) * n), dtype=np.uint8).copy()


def pack(t, p):
    sd, cam = t.scene_desc(), t.camera_desc()
    return dict(spheres=raw(sd.spheres, capi.Sphere, sd.n_spheres), materials=raw(sd.materials, capi.Material, sd.n_materials),
                textures=raw(sd.textures, capi.Texture, sd.n_textures),
                camera=np.frombuffer(bytes(cam), dtype=np.uint8).copy(), params=np.frombuffer(bytes(p), dtype=np.uint8).copy(),
                rng_state=t.rng_state())


def case(name, t, spp, bounces, render_seed, precisions=(0, 1), mode_a=True, **gpu):
    t.samples_per_px, t.max_bounces = spp, bounces
    t.set_gpu(render_seed=render_seed, **gpu)
    sd, cam = t.scene_desc(), t.camera_desc()
    out = {}
    for prec in precisions:
        t.set_gpu(precision=prec)
        p = t.params()
        img, st = oracle.render_b(sd, cam, p)
        tag = "f64" if prec else "f32"
        out[f"image_b_{tag}"] = img
        out[f"segments_b_{tag}"] = np.uint64(st.segments)
        out[f"params_{tag}"] = np.frombuffer(bytes(p), dtype=np.uint8).copy()
    t.set_gpu(precision=0)
    out.update(pack(t, t.params()))
    if mode_a:
        pa = t.params()
        pa.precision, pa.tmin = 1, 1e-10
        rs = t.rng_state().copy()
        img, st = oracle.render_a(sd, cam, pa, rs)
        out["image_a"] = img
        out["segments_a"] = np.uint64(st.segments)
        rs = t.rng_state().copy()
        img, st = oracle.render_a(sd, cam, pa, rs, linear=True)
        out["image_a_linear"] = img
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("image")})


def main():
    oracle.build()
    case("three_spheres_64x36_8spp", tracer.threeSpheres(64, seed=1), 8, 50, 1)
    case("random_bouncing_48x27_4spp", tracer.randomBouncing(48, seed=42), 4, 50, 5)
    case("random_bouncing_grid3_32x18_6spp_chunk4", tracer.randomBouncing(32, -3, 3, seed=9), 6, 8, 77, chunk_spp=4)


if __name__ == "__main__":
    main()
