"""The f32 reject filter is conservative: over the segments of BASELINE's 10k-sphere scene, no
sphere whose f64 discriminant is >= 0 is filtered out (reference semantics src/geom.zig:43-50).
The audit replays paths over the flat list in the oracle's mode B — the arithmetic the GPU reproduces bit for bit —
and evaluates the f64 discriminant for EVERY (segment, sphere) pair."""
import numpy as np

from rayz_amd import capi, tracer


def test_no_false_negatives_over_config3_segments(oracle):
    t = tracer.randomBouncing(1920, -50, 50, seed=42)  # configs[2]'s scene and camera
    t.samples_per_px, t.max_bounces = 2, 50
    t.set_gpu(render_seed=1, traversal=capi.TRAVERSAL_LINEAR)
    sd, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    rng = np.random.default_rng(0)
    rows = rng.integers(0, p.height, 6000)
    pixels = rows * p.width + rng.integers(0, p.width, 6000)  # spread over the frame: sky, spheres, ground, horizon
    au = oracle.filter_audit(sd, cam, p, pixels, capi.PRECISION_F32)
    assert au["pairs"] > 2.5e8  # ~3 segments x 10,003 spheres x 12,000 paths
    assert au["false_negatives"] == 0
    assert au["f64_hits"] <= au["candidates"] <= 1.25 * au["f64_hits"] + 1000  # the padding lets few extra pairs through
    print("filter audit f32:", au)
    # f64 fidelity mode: the flat list's reject test is the SAME f32 test, on the f64 ray narrowed to f32 (pad 40u): it
    # must let through every pair the f64 quadratic of the f64 ray hits
    au64 = oracle.filter_audit(sd, cam, p, pixels, capi.PRECISION_F64)
    assert au64["false_negatives"] == 0 and au64["pairs"] > 2.5e8
    assert au64["f64_hits"] <= au64["candidates"] <= 1.25 * au64["f64_hits"] + 1000
    print("filter audit f64 rays through the f32 filter:", au64)


def test_no_false_negatives_on_grazing_rays_and_big_coordinates(oracle):
    """Adversarial pairs instead of typical ones: rays grazing the r = 1000 ground from points on it, and a scene
    shifted far from the origin (the slack scales with |c| + |o|)."""
    import kat_records as K

    rng = np.random.default_rng(9)
    for shift in (0.0, 3.0e4):
        rec = K.random_sphere_hits(rng, 300_000, big=True)
        rec[:, 0:3] = K.f32r(rec[:, 0:3] + shift)
        rec[:, 7:10] = K.f32r(rec[:, 7:10] + shift)
        a = oracle.kat_a(capi.KAT_SPHERE_HIT, rec)
        for prec in (capi.PRECISION_F32, capi.PRECISION_F64):
            b = oracle.kat_b(capi.KAT_SPHERE_HIT, rec, prec)
            hit = a[:, 0] == 1
            assert hit.sum() > 50_000 and (b[hit, 9] == 1).all(), (shift, prec, int((b[hit, 9] == 0).sum()))
