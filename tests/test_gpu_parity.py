"""GPU (HIP, through the C ABI) versus the oracle's mode B: bit-exact, f32 and f64.

Mode B is the arithmetic the kernel is specified to perform; north_star's tolerance is
1e-4 per channel against the seeded CPU image — these tests hold the kernel to exact equality instead.
"""
import numpy as np
import pytest

from rayz_amd import capi, tracer

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: per-pixel RGB within 1e-4 of the seeded CPU image (we assert equality, then this)


def _assert_same(got, want, what):
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
        pytest.fail(f"{what}: {len(bad)} of {got.size} values differ, max |d| {err:.3e} "
                    f"(tolerance {TOL}); first at {bad[:5].tolist()}")


def _render_pair(gpu, oracle, t, **param_overrides):
    scene, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    for k, v in param_overrides.items():
        setattr(p, k, v)
    got, gst = gpu.render_host(scene, cam, p)
    want, ost = oracle.render_b(scene, cam, p)
    return got, want, gst, ost


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_config1_three_spheres_400x225_8spp(gpu, oracle, seed):
    """BASELINE.json configs[0]: 3 Lambertian spheres, 400x225, 8 spp."""
    t = tracer.threeSpheres(400, seed=seed)
    t.samples_per_px = 8
    t.set_gpu(render_seed=seed)
    got, want, gst, ost = _render_pair(gpu, oracle, t)
    _assert_same(got, want, f"config 1 seed {seed}")
    assert gst.primary_rays == 400 * 225 * 8 == ost.primary_rays
    assert gst.segments == ost.segments
    assert gst.sphere_tests == ost.sphere_tests == ost.segments * 3


def test_random_bouncing_all_materials(gpu, oracle):
    """configs[1]'s scene (~485 spheres: checker ground, glass, metal+fuzz, moving diffuse) at test size."""
    t = tracer.randomBouncing(160, seed=42)
    t.samples_per_px = 16
    t.set_gpu(render_seed=5)
    got, want, gst, ost = _render_pair(gpu, oracle, t)
    _assert_same(got, want, "randomBouncing 160x90x16")
    assert gst.segments == ost.segments


def test_random_bouncing_f64(gpu, oracle):
    t = tracer.randomBouncing(96, seed=3)
    t.samples_per_px = 8
    t.set_gpu(render_seed=9, precision=capi.PRECISION_F64)
    got, want, gst, ost = _render_pair(gpu, oracle, t)
    assert got.dtype == np.float64
    _assert_same(got, want, "randomBouncing f64")
    assert gst.segments == ost.segments
