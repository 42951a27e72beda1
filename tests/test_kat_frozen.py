"""The functions the reference's own tests do NOT pin — `hitInner` (src/geom.zig:38-66), `scatter`
(src/material.zig:73-160), `reflectance` / `refract` (:179-194), `getRay` (src/camera.zig:59-90), `AABB.hit`
(src/hit.zig:70-98), the checker, the background — frozen: tests/golden/kat_frozen.npz holds a fixed record set with the
outputs of mode A (the line-by-line restatement of the reference) and of mode B (the kernels' arithmetic, f32 and f64)
as of the commit that generated it (tests/golden/make_kat_golden.py).  Mode A, mode B and the HIP device functions are
held to it EXACTLY, so an edit that moves the oracle and the kernel together no longer passes silently."""
import os

import numpy as np
import pytest

from rayz_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
OPS = {"refract": capi.KAT_REFRACT, "reflectance": capi.KAT_REFLECTANCE, "get_ray": capi.KAT_GET_RAY, "box_hit": capi.KAT_BOX_HIT,
       "sphere_hit": capi.KAT_SPHERE_HIT, "scatter": capi.KAT_SCATTER, "checker": capi.KAT_CHECKER, "background": capi.KAT_BACKGROUND,
       "triangle_hit": capi.KAT_TRIANGLE_HIT, "scan_discs": capi.KAT_SCAN_DISCS}


@pytest.fixture(scope="module")
def frozen():
    return np.load(os.path.join(HERE, "golden", "kat_frozen.npz"))


def _same(got, want):
    return (got == want) | (np.isnan(got) & np.isnan(want))


@pytest.mark.parametrize("name", sorted(OPS))
def test_mode_a_reproduces_the_frozen_outputs(oracle, frozen, name):
    if name == "triangle_hit":
        pytest.skip("build-defined primitive: the reference has no triangle, nothing of it to freeze")
    got, want = oracle.kat_a(OPS[name], frozen["in_" + name]), frozen["a_" + name]
    if name == "reflectance":  # mode A calls libm's pow(x, 5) as the reference does (src/material.zig:182): a few ulps of libm
        assert np.allclose(got, want, rtol=1e-15, atol=0)
        return
    bad = ~_same(got, want).all(1)
    assert not bad.any(), (name, int(bad.sum()), np.flatnonzero(bad)[:5].tolist())


@pytest.mark.parametrize("prec,tag", [(capi.PRECISION_F32, "b32_"), (capi.PRECISION_F64, "b64_")])
@pytest.mark.parametrize("name", sorted(OPS))
def test_mode_b_reproduces_the_frozen_outputs(oracle, frozen, name, prec, tag):
    got, want = oracle.kat_b(OPS[name], frozen["in_" + name], prec), frozen[tag + name]
    bad = ~_same(got, want).all(1)
    assert not bad.any(), (name, tag, int(bad.sum()), np.flatnonzero(bad)[:5].tolist())


def test_frozen_mode_b_agrees_with_frozen_mode_a(frozen):
    """The frozen values themselves: mode B's decisions equal mode A's except at rounding borders, values within the
    bounds tests/test_kat_cpu.py derives (spot form: loose constants, no margins analysis)."""
    a, b64, b32 = frozen["a_sphere_hit"], frozen["b64_sphere_hit"], frozen["b32_sphere_hit"]
    assert (a[:, 0] == b64[:, 0]).mean() > 0.999 and (a[:, 0] == b32[:, 0]).mean() > 0.995
    assert (b64[a[:, 0] == 1, 9] == 1).all() and (b32[a[:, 0] == 1, 9] == 1).all()  # the filter never drops a reference hit
    both = (a[:, 0] == 1) & (b64[:, 0] == 1) & (b32[:, 0] == 1)
    assert 800 < both.sum() < 2600
    assert np.abs(a[both, 1] - b64[both, 1]).max() < 1e-6 and np.abs(a[both, 1] / b32[both, 1] - 1).max() < 1e-3
    a, b64, b32 = frozen["a_scatter"], frozen["b64_scatter"], frozen["b32_scatter"]
    assert (a[:, 0] == b64[:, 0]).mean() > 0.999 and (a[:, 4] == b64[:, 4]).mean() > 0.999  # scattered?, number of draws
    ok = (a[:, 0] == 1) & (b64[:, 0] == 1) & (a[:, 4] == b64[:, 4])
    assert np.abs(a[ok, 1:4] - b64[ok, 1:4]).max() < 1e-9
    for k in ("refract", "get_ray", "background"):
        assert np.nanmax(np.abs(frozen["a_" + k][:, :6] - frozen["b64_" + k][:, :6])) < 1e-9, k
        assert np.nanmax(np.abs(frozen["a_" + k][:, :6] - frozen["b32_" + k][:, :6])) < 2e-4 * (1 + np.abs(frozen["a_" + k][:, :6]).max()), k
    # boxes: mode B's test is conservative — it never misses a box the reference's strict test hits
    assert (frozen["b32_box_hit"][frozen["a_box_hit"][:, 0] == 1, 0] == 1).all() and (frozen["b64_box_hit"][frozen["a_box_hit"][:, 0] == 1, 0] == 1).all()
    assert (frozen["a_checker"][:, 0] == frozen["b64_checker"][:, 0]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tag", [(capi.PRECISION_F32, "b32_"), (capi.PRECISION_F64, "b64_")])
def test_device_functions_reproduce_the_frozen_mode_b_outputs(gpu, frozen, prec, tag):
    """rayz_hip_kat (the trace kernels' own inlined device functions) against the FROZEN file, not against today's oracle."""
    for name, op in sorted(OPS.items()):
        got, want = gpu.kat(op, frozen["in_" + name], prec), frozen[tag + name]
        bad = ~_same(got, want).all(1)
        assert not bad.any(), (name, tag, int(bad.sum()), np.flatnonzero(bad)[:5].tolist())
