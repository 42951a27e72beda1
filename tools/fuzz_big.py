#!/usr/bin/env python3
"""Random scenes with hundreds to thousands of hittables (trees deeper than the LDS copy of their top, oversized
hittables peeled or not, spheres + triangles): flat list == BVH exactly on the device, f32 and f64; every 8th scene also
against the oracle.  usage: fuzz_big.py <first_seed> <count>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import binding as oracle
from rayz_amd import capi, render, tracer

render.init(0); oracle.load()
if 'nodes16' in sys.argv:  # force the 32-byte node records (16-bit plane indices) however small the tree
    sys.argv.remove('nodes16'); render.debug_set(capi.DEBUG_BVH_NODES, 2)


def big_scene(seed):
    rng = np.random.default_rng(seed)
    n_sph, n_tri = int(rng.integers(200, 3000)), int(rng.choice([0, 0, 200, 1500]))
    ext = float(rng.choice([6.0, 20.0, 60.0]))
    t = tracer.Tracer.init(int(rng.integers(48, 96)), float(rng.uniform(20, 60)), float(rng.uniform(4, 30)),
                           float(rng.choice([0.0, 0.5])), rng.uniform(-ext, ext, 3) * [1, 0.3, 1] + [0, 2 + ext * 0.2, 0],
                           rng.uniform(-1, 1, 3), (0, 1, 0), seed=seed)
    P = t.pool
    tex = [P.add_solid_texture(rng.uniform(0.05, 0.95, 3)) for _ in range(3)]
    tex.append(P.add_checker_texture(0.7, tex[0], tex[1]))
    mats = [P.add_diffuse(int(rng.choice(tex)), 2), P.add_diffuse(int(rng.choice(tex)), 0), P.add_metallic(tex[0], 0.0),
            P.add_metallic(tex[1], 0.4), P.add_dielectric(1.5)]
    if rng.random() < 0.7:
        P.add_sphere((0, -1000, 0), 1000.0, int(rng.choice(mats)))  # an oversized hittable (peeled out of the device tree)
    for _ in range(n_sph):
        c = rng.uniform(-ext, ext, 3)
        c[1] = abs(c[1]) * 0.15 + 0.2
        v = (0, 0, 0) if rng.random() < 0.5 else ((0, float(rng.uniform(0, 0.5)), 0) if rng.random() < 0.8 else tuple(rng.uniform(-0.3, 0.3, 3)))
        P.add_sphere(c, float(rng.uniform(0.05, 0.4)), int(rng.choice(mats)), velocity=v)
    for _ in range(n_tri):
        b = rng.uniform(-ext, ext, 3)
        b[1] = abs(b[1]) * 0.1
        P.add_triangle(b, b + rng.uniform(-0.8, 0.8, 3), b + rng.uniform(-0.8, 0.8, 3), int(rng.choice(mats)))
    t.samples_per_px, t.max_bounces = int(rng.integers(2, 12)), int(rng.integers(3, 30))
    t.set_gpu(render_seed=int(rng.integers(0, 2 ** 62)))
    return t


first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    t = big_scene(seed)
    for prec in (capi.PRECISION_F32, capi.PRECISION_F64):
        got = {}
        for trav in (capi.TRAVERSAL_LINEAR, capi.TRAVERSAL_BVH):
            t.set_gpu(traversal=trav, precision=prec)
            got[trav] = render.render_host(t.scene_desc(), t.camera_desc(), t.params())
        a, b = got[capi.TRAVERSAL_LINEAR], got[capi.TRAVERSAL_BVH]
        if not (np.array_equal(a[0], b[0], equal_nan=True) and a[1].segments == b[1].segments):
            bad += 1
            print(f"MISMATCH flat vs bvh: seed {seed} precision {prec}: {int((a[0] != b[0]).sum())} values, segments {a[1].segments} vs {b[1].segments}", flush=True)
        if seed % 8 == 0:
            want, ost = oracle.render_b(t.scene_desc(), t.camera_desc(), t.params())
            if not (np.array_equal(b[0], want, equal_nan=True) and b[1].segments == ost.segments):
                bad += 1
                print(f"MISMATCH gpu bvh vs oracle: seed {seed} precision {prec}", flush=True)
    if (seed - first) % 25 == 24:
        print(f"... {seed - first + 1} scenes, {bad} mismatches", flush=True)
print(f"done: {count} scenes, {bad} mismatches")
