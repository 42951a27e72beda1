"""Several GPUs behind ONE call (`rayz_hip_multi_*`, `rayz_hip_render_multi`): the single `tracer.render()` of the
reference's caller (src/rayz.zig:26) driving every device of the node.  The GPU box has one MI355X, so what runs
here is (a) the degenerate n = 1 case through the complete machinery — per-device contexts, ncclCommInitAll +
ncclGather (or peer copies), the un-interleave kernel — and (b) n = 2, 3, 8 with every "device" being device 0
(the RAYZ_GATHER_ALLOW_DUPLICATE_DEVICES transport flag, peer copies only: RCCL refuses a device twice): the N-way row dealing, N scenes,
the gather into N slots and the un-interleave run for real.  Both must be bit-identical to the single-device entry
points."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_images_equal
from rayz_amd import capi, render, tracer

pytestmark = pytest.mark.gpu


def _scene():
    t = tracer.randomBouncing(96, -4, 4, seed=21)
    t.samples_per_px, t.max_bounces = 6, 10
    t.set_gpu(render_seed=3)
    return t


@pytest.mark.parametrize("transport", [capi.GATHER_RCCL, capi.GATHER_PEER_COPY])
@pytest.mark.parametrize("tile_rows", [0, 1, 8])
def test_multi_n1_is_bit_identical_to_single_device(gpu, oracle, transport, tile_rows):
    t = _scene()
    sd, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    want, wst = gpu.render_host(sd, cam, p)
    ref, _ = oracle.render_b(sd, cam, p)
    assert_images_equal(want, ref, "single device vs oracle")
    p.tile_rows = tile_rows
    m = render.MultiScene(sd, [0], transport)
    info = m.info()
    assert info["n_devices"] == 1 and info["transport"] == transport
    assert (info["rccl_version"] > 0) == (transport == capi.GATHER_RCCL)
    for _ in range(2):  # the handle is reusable
        got, st = m.render(cam, p)
        assert_images_equal(got, want, f"multi n=1 transport {transport} tile_rows {tile_rows}")
        assert (st.primary_rays, st.segments) == (wst.primary_rays, wst.segments)
        assert st.kernel_ms > 0
    m.close()


@pytest.mark.parametrize("n", [2, 3, 8])
@pytest.mark.parametrize("tile_rows", [0, 5])
def test_multi_n_way_on_one_device(gpu, oracle, n, tile_rows):
    """N shards through rayz_hip_multi_render on a one-GPU box: the same device listed N times (test-only flag)."""
    t = _scene()
    sd, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    want, wst = gpu.render_host(sd, cam, p)
    with pytest.raises(capi.RayzHipError, match="listed twice"):
        render.MultiScene(sd, [0] * n, capi.GATHER_PEER_COPY)
    with pytest.raises(capi.RayzHipError, match="needs the peer-copy transport"):  # never through RCCL
        render.MultiScene(sd, [0] * n, capi.GATHER_RCCL | capi.GATHER_ALLOW_DUPLICATE_DEVICES)
    m = render.MultiScene(sd, [0] * n, capi.GATHER_PEER_COPY | capi.GATHER_ALLOW_DUPLICATE_DEVICES)
    assert m.info()["n_devices"] == n and m.info()["transport"] == capi.GATHER_PEER_COPY
    with pytest.raises(capi.RayzHipError, match="no frame"):
        m.device_stats()
    p.tile_rows = tile_rows
    for _ in range(2):
        got, st = m.render(cam, p)
        assert_images_equal(got, want, f"multi n={n} (one device) tile_rows {tile_rows}")
        assert (st.primary_rays, st.segments) == (wst.primary_rays, wst.segments)
        # per-device counters: every shard's own rows / segments / kernel time; they add up to the frame's
        per = m.device_stats()
        assert len(per) == n and sum(d.primary_rays for d in per) == st.primary_rays
        # (a device the deal leaves without rows — 8-row tiles of a small frame on 8 devices — launches nothing)
        assert sum(d.segments for d in per) == st.segments and all((d.kernel_ms > 0) == (d.primary_rays > 0) for d in per)
        assert max(d.kernel_ms for d in per) == st.kernel_ms
        gather_ms, frame_ms = m.timing()
        assert 0 < gather_ms < 1e4 and frame_ms >= gather_ms * 0.5
    u8, _ = m.render(cam, p, u8=True)
    img = tracer.Image(p.height, p.width)
    img.pixels = want.astype(np.float64)
    assert np.array_equal(u8, img.to_u8())
    t.set_gpu(precision=capi.PRECISION_F64, traversal=capi.TRAVERSAL_BVH)
    p64 = t.params()
    p64.tile_rows = tile_rows
    got64, _ = m.render(cam, p64)
    want64, _ = oracle.render_b(sd, cam, p64)
    assert_images_equal(got64, want64, f"multi n={n} f64 BVH")
    m.close()


def test_multi_f64_and_u8(gpu, oracle):
    t = _scene()
    sd, cam = t.scene_desc(), t.camera_desc()
    m = render.MultiScene(sd, [0])
    p = t.params()
    f32, _ = m.render(cam, p)
    u8, _ = m.render(cam, p, u8=True)  # writePPM's transform on the device, before the gather
    img = tracer.Image(p.height, p.width)
    img.pixels = f32.astype(np.float64)
    assert u8.dtype == np.uint8 and np.array_equal(u8, img.to_u8())
    t.set_gpu(precision=capi.PRECISION_F64)
    p64 = t.params()
    got, _ = m.render(cam, p64)
    want, _ = oracle.render_b(sd, cam, p64)
    assert_images_equal(got, want, "multi f64")
    # growing frame on the same handle (buffers are grow-only)
    p.width, p.height = 128, 72
    big, _ = m.render(cam, p)
    want, _ = gpu.render_host(sd, cam, p)
    assert_images_equal(big, want, "multi after growing the frame")
    m.close()


def test_multi_argument_errors_on_a_live_device(gpu):
    lib = capi.load()
    t = _scene()
    sd, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    m = render.MultiScene(sd, [0])
    out = np.zeros((p.height, p.width, 3), dtype=np.float32)
    q = capi.RenderParams.from_buffer_copy(bytes(p))
    q.shard_index, q.shard_count = 1, 2
    rc = lib.rayz_hip_multi_render(m._h, C.byref(cam), C.byref(q), out.ctypes.data_as(C.c_void_p), None)
    assert rc == capi.ERR_BAD_ARG and b"shards the frame itself" in lib.rayz_hip_last_error()
    q = capi.RenderParams.from_buffer_copy(bytes(p))
    q.precision = capi.PRECISION_F64
    rc = lib.rayz_hip_multi_render(m._h, C.byref(cam), C.byref(q), out.ctypes.data_as(C.c_void_p), None)
    assert rc == capi.ERR_BAD_ARG
    m.close()
    with pytest.raises(capi.RayzHipError, match="out of range"):
        render.MultiScene(sd, [63])  # no such device on this box


def test_tracer_render_on_a_device_list(gpu, oracle):
    """The host mirror's `Tracer.render()` with gpu.devices set goes through rayz_hip_render_multi (one-shot form)."""
    t = _scene()
    want, _ = oracle.render_b(t.scene_desc(), t.camera_desc(), t.params())
    t.set_gpu(devices=[0])
    for k in range(3):  # the Tracer keeps ONE RayzMulti (scenes + communicator) across frames: created by the first call only
        rays = t.render()
        assert rays == t.info().width * t.info().height * 6
        assert_images_equal(t.img.pixels.astype(np.float32), want, f"Tracer.render on devices [0], frame {k}")


def test_scene_bound_to_a_device_and_foreign_current_device(gpu, oracle):
    """A scene created with rayz_hip_scene_create_on keeps its device; the library selects it itself, whatever the
    calling thread's current device is (here: a second host thread, whose HIP current device was never set)."""
    import threading

    t = _scene()
    sd, cam, p = t.scene_desc(), t.camera_desc(), t.params()
    want, _ = oracle.render_b(sd, cam, p)
    res = {}

    def work():
        import torch

        ds = render.DeviceScene(sd, device=0)
        out = torch.empty((p.height, p.width, 3), dtype=torch.float32, device="cuda:0")
        ds.render_into(cam, p, out.data_ptr(), 0)
        ds.sync()
        res["img"] = out.cpu().numpy()
        ds.close()

    th = threading.Thread(target=work)
    th.start()
    th.join()
    assert_images_equal(res["img"], want, "render from a second host thread")
