"""Randomised scenes: every hittable class (static / y-moving / generally moving spheres, triangles), every
material and texture kind, random cameras (with and without defocus) — GPU vs oracle mode B, bit for bit, in both
precisions and both traversals."""
import numpy as np
import pytest

from helpers import assert_images_equal
from rayz_amd import capi, tracer

pytestmark = pytest.mark.gpu


def random_scene(seed):
    rng = np.random.default_rng(seed)
    look_from = rng.uniform(-6, 6, 3) + np.array([0, 3.0, 0])
    t = tracer.Tracer.init(int(rng.integers(24, 72)), float(rng.uniform(20, 70)), float(rng.uniform(2, 10)),
                           float(rng.choice([0.0, 0.0, 1.0, 3.0])), look_from, rng.uniform(-1, 1, 3), (0, 1, 0), seed=seed)
    P = t.pool
    tex = [P.add_solid_texture(rng.uniform(0.05, 0.95, 3)) for _ in range(4)]
    tex.append(P.add_checker_texture(float(rng.uniform(0.1, 1.0)), tex[0], tex[1]))
    tex.append(P.add_checker_texture(float(rng.uniform(0.3, 2.0)), tex[4], tex[2]))
    mats = []
    for k in range(10):
        kind = rng.integers(0, 3)
        if kind == 0:
            mats.append(P.add_diffuse(int(rng.choice(tex)), int(rng.integers(0, 3))))
        elif kind == 1:
            mats.append(P.add_metallic(int(rng.choice(tex)), float(rng.choice([0.0, rng.uniform(0, 1.5)]))))
        else:
            mats.append(P.add_dielectric(float(rng.choice([1.5, 1.33, 1 / 1.5, 2.4]))))
    P.add_sphere((0, -500, 0), 500.0, mats[0])
    for _ in range(int(rng.integers(0, 40))):
        cls = rng.integers(0, 3)
        v = (0, 0, 0) if cls == 0 else ((0, float(rng.uniform(-1, 1)), 0) if cls == 1 else tuple(rng.uniform(-1, 1, 3)))
        c = rng.uniform(-4, 4, 3)
        c[1] = abs(c[1]) * 0.5 + 0.2
        P.add_sphere(c, float(rng.uniform(0.1, 0.9)), int(rng.choice(mats)), velocity=v)
    for _ in range(int(rng.integers(0, 30))):
        base = rng.uniform(-4, 4, 3)
        base[1] = abs(base[1]) * 0.5
        P.add_triangle(base, base + rng.uniform(-1.5, 1.5, 3), base + rng.uniform(-1.5, 1.5, 3), int(rng.choice(mats)))
    t.samples_per_px = int(rng.integers(1, 20))
    t.max_bounces = int(rng.integers(1, 30))
    t.set_gpu(render_seed=int(rng.integers(0, 2 ** 62)), chunk_spp=int(rng.choice([0, 1, 3, 16])))
    return t


# 1000..1011, and three scenes on which the BVH walk once lost hits to a zero direction component (tests/test_kat_cpu.py)
@pytest.mark.parametrize("seed", list(range(1000, 1012)) + [5003, 5008, 5010])
def test_random_scene_parity(gpu, oracle, seed):
    t = random_scene(seed)
    for trav in (capi.TRAVERSAL_LINEAR, capi.TRAVERSAL_BVH):
        for prec in (capi.PRECISION_F32, capi.PRECISION_F64):
            t.set_gpu(traversal=trav, precision=prec)
            scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
            got, gst = gpu.render_host(scene, cam, p)
            want, ost = oracle.render_b(scene, cam, p)
            assert_images_equal(got, want, f"seed {seed} traversal {trav} precision {prec}")
            assert gst.segments == ost.segments
