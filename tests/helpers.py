"""Shared helpers for the tests: golden-fixture loading and image comparison."""
import ctypes as C
import os

import numpy as np

from rayz_amd import capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ["three_spheres_64x36_8spp", "random_bouncing_48x27_4spp", "random_bouncing_grid3_32x18_6spp_chunk4"]


class Golden:
    """A committed fixture: the flattened pool + camera + params as raw ABI bytes, and the oracle's images."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

        def arr(key, T):
            raw = self.z[key].tobytes()
            n = len(raw) // C.sizeof(T)
            a = (T * n).from_buffer_copy(raw)
            return a, n

        self.spheres, ns = arr("spheres", capi.Sphere)
        self.materials, nm = arr("materials", capi.Material)
        self.textures, nt = arr("textures", capi.Texture)
        self.scene = capi.SceneDesc(spheres=self.spheres, materials=self.materials, textures=self.textures,
                                    n_spheres=ns, n_materials=nm, n_textures=nt)
        self.camera = capi.CameraDesc.from_buffer_copy(self.z["camera"].tobytes())
        self.rng_state = self.z["rng_state"].copy()

    def params(self, tag="f32"):
        return capi.RenderParams.from_buffer_copy(self.z[f"params_{tag}"].tobytes())

    def image(self, which):
        return self.z[which]


def assert_images_equal(got, want, what):
    assert got.shape == want.shape and got.dtype == want.dtype, (what, got.shape, want.shape, got.dtype, want.dtype)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
        raise AssertionError(f"{what}: {len(bad)} of {got.size} values differ, max |d| {err:.3e}; first at "
                             f"{bad[:5].tolist()}")
