#!/usr/bin/env python3
"""Regenerate tests/golden/kat_frozen.npz: a FIXED set of known-answer records (sphere hits incl. the r = 1000 grazing
regime, scatter events with their uniforms, camera rays, boxes incl. zero direction components, refraction, Schlick,
checker cells, background, triangles) together with what this repository's oracle computes for them TODAY:

  a_<op>    mode A — the line-by-line f64 restatement of the reference's own functions
            (src/geom.zig:38-66 hitInner, src/material.zig:73-160 scatter, :179-194, src/camera.zig:59-90,
            src/hit.zig:70-98, src/renderer.zig:124-125) — which the reference's own tests do not pin;
  b32_<op>, b64_<op>   mode B — the kernels' arithmetic, f32 and f64.

The reference (Zig) cannot run here, so these are not outputs OF the reference; they freeze the restatement, so that an
edit which moves oracle and kernel TOGETHER can no longer pass unnoticed: tests/test_kat_frozen.py holds mode A and
mode B to this file exactly (CPU) and the HIP device functions to the frozen mode-B values exactly (GPU).
Regenerate ONLY for a deliberate change of the arithmetic contract, and say so in the commit.

    python tests/golden/make_kat_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import kat_records as K  # noqa: E402
from oracle import binding as oracle  # noqa: E402
from rayz_amd import capi  # noqa: E402

OPS = {"refract": capi.KAT_REFRACT, "reflectance": capi.KAT_REFLECTANCE, "get_ray": capi.KAT_GET_RAY, "box_hit": capi.KAT_BOX_HIT,
       "sphere_hit": capi.KAT_SPHERE_HIT, "scatter": capi.KAT_SCATTER, "checker": capi.KAT_CHECKER, "background": capi.KAT_BACKGROUND,
       "triangle_hit": capi.KAT_TRIANGLE_HIT, "scan_discs": capi.KAT_SCAN_DISCS}


def records():
    rng = np.random.default_rng(20261004)
    cam = capi.CameraDesc()  # randomBouncing's camera (src/rayz.zig:46-55) at 1920x1080
    oracle.load().rayz_oracle_camera_init(20, 10.0, 0.6, oracle.d3([13, 2, 3]), oracle.d3([0, 0, 0]), oracle.d3([0, 1, 0]), 1080, 1920, cam)
    refl = K.blank(400)
    refl[:, 0], refl[:, 1] = rng.uniform(0, 1, 400), K.f32r(rng.uniform(0.4, 2.5, 400))
    refl[:, 0] = K.f32r(refl[:, 0])
    bg = K.blank(400)
    bg[:, 0:3] = K.f32r(rng.normal(size=(400, 3)))
    return {
        "refract": np.concatenate([K.refract_reference()[0], K.random_refracts(rng, 600)]),
        "reflectance": refl,
        "get_ray": K.random_get_rays(rng, 800, cam),
        "box_hit": np.concatenate([K.box_hit_reference()[0], K.random_boxes(rng, 1200), K.axis_parallel_boxes(rng, 800)[0]]),
        "sphere_hit": np.concatenate([K.random_sphere_hits(rng, 1500), K.random_sphere_hits(rng, 1500, big=True)]),
        "scatter": K.random_scatters(rng, 2500),
        "checker": K.random_checkers(rng, 600),
        "background": bg,
        "triangle_hit": K.random_triangles(rng, 1200),
        "scan_discs": K.random_scan_blocks(rng, 1500),  # LAST: the draws of the sets above stay what they were
    }


def main():
    oracle.build()
    out = {}
    for name, rec in records().items():
        op = OPS[name]
        out["in_" + name] = rec
        if name != "triangle_hit":  # the triangle is build-defined: the reference has none, mode A's Möller–Trumbore is ours too
            out["a_" + name] = oracle.kat_a(op, rec)
        out["b32_" + name] = oracle.kat_b(op, rec, capi.PRECISION_F32)
        out["b64_" + name] = oracle.kat_b(op, rec, capi.PRECISION_F64)
    path = os.path.join(HERE, "kat_frozen.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items() if k.startswith("in_")})


if __name__ == "__main__":
    main()
