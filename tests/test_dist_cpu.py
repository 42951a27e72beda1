"""The N > 1 path on CPU: two gloo ranks shard a frame by interleaved row tiles, each "renders" its rows (with
the oracle standing in for the GPU, test-side only), one all_gather reassembles the frame, and the result is
identical to the unsharded image — and hashes to the same `frame_sha256` bench.py prints for N = 1.  Exercises
rayz_amd/dist.py exactly as bench.py uses it."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker_u8(rank, world, port, q):
    """The u8 form of the gather: each rank tone-maps ITS rows before the collective (on the GPU:
    rayz_hip_tonemap_u8; here the host mirror's Image.to_u8 stands in), the gathered frame equals Image.to_u8()."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as oracle
        from rayz_amd import dist as rdist
        from rayz_amd import tracer

        t = tracer.randomBouncing(40, -2, 2, seed=5)
        t.samples_per_px, t.max_bounces = 3, 6
        t.set_gpu(render_seed=8)
        sd, cam, p = t.scene_desc(), t.camera_desc(), t.params()

        def to_u8(src, dst):
            img = tracer.Image(src.shape[0], src.shape[1])
            img.pixels = src.numpy().astype(np.float64)
            dst.copy_(torch.from_numpy(img.to_u8()))

        fg = rdist.FrameGather(p.height, p.width, world, rank, torch.device("cpu"), dtype=torch.uint8, tonemap=to_u8)
        mine, _ = oracle.render_b(sd, cam, rdist.shard_params(p, rank, world), threads=1)
        fg.tile[: mine.shape[0]] = torch.from_numpy(mine)
        frame = fg.gather()
        assert frame.dtype == torch.uint8 and fg._gathered.dtype == torch.uint8  # u8 is what travelled
        if rank == 0:
            full, _ = oracle.render_b(sd, cam, p, threads=1)
            img = tracer.Image(p.height, p.width)
            img.pixels = full.astype(np.float64)
            q.put(bool(np.array_equal(frame.numpy(), img.to_u8())))
    finally:
        dist.destroy_process_group()


def test_u8_gather_equals_image_to_u8_gloo(built):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29500 + (os.getpid() * 11 + 5) % 2000
    mp.spawn(_worker_u8, args=(2, port, q), nprocs=2, join=True)
    assert q.get() is True


def _worker(rank, world, port, tile_rows, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as oracle
        from rayz_amd import dist as rdist
        from rayz_amd import tracer

        t = tracer.randomBouncing(40, -2, 2, seed=5)
        t.samples_per_px, t.max_bounces = 3, 6
        t.set_gpu(render_seed=8)
        sd, cam, p = t.scene_desc(), t.camera_desc(), t.params()
        if tile_rows == 0:  # the default deal, exactly as bench.py makes it
            fg = rdist.FrameGather(p.height, p.width, world, rank, torch.device("cpu"))
            mine, _ = oracle.render_b(sd, cam, rdist.shard_params(p, rank, world), threads=1)
        else:
            fg = rdist.FrameGather(p.height, p.width, world, rank, torch.device("cpu"), tile_rows=tile_rows)
            mine, _ = oracle.render_b(sd, cam, rdist.shard_params(p, rank, world, tile_rows), threads=1)
        assert mine.shape[0] == len(fg.my_rows)
        fg.tile[: mine.shape[0]] = torch.from_numpy(mine)
        frame_t = fg.gather()
        sha = rdist.frame_sha256(frame_t)  # what bench.py prints as `frame_sha256`
        frame = frame_t.numpy().copy()
        if rank == 0:
            full, _ = oracle.render_b(sd, cam, p, threads=1)
            one = rdist.FrameGather(p.height, p.width, 1, 0, torch.device("cpu"))  # the N = 1 line's frame
            one.tile.copy_(torch.from_numpy(full))
            q.put(bool(np.array_equal(frame, full)) and sha == rdist.frame_sha256(one.gather()) and len(sha) == 64)
        # the per-rank figures of bench.py's N > 1 line: every rank gets every rank's row
        table = rdist.rank_stats([10.0 + rank, 0.5 * (rank + 1), 1000.0 * (rank + 1)], torch.device("cpu"), world)
        assert table.shape == (world, 3) and table[:, 0].tolist() == [10.0 + r for r in range(world)]
        blk = rdist.per_rank_block(table)
        assert blk["kernel_ms"]["max"] == 10.0 + world - 1 and blk["kernel_ms"]["min"] == 10.0 and len(blk["kernel_ms"]["all"]) == world
        assert blk["gather_ms"]["mean"] == 0.5 * (world + 1) / 2 and blk["segments"] == [1000.0 * (r + 1) for r in range(world)]
        # every rank holds the whole frame
        chk = torch.tensor([float(frame.sum())], dtype=torch.float64)
        lst = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        assert all(float(x) == float(lst[0]) for x in lst)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_rows", [(2, 0), (2, 1), (2, 8), (3, 4), (3, 0)])  # 0 = the default deal (8-row tiles)
def test_row_tile_shard_and_gather_gloo(built, world, tile_rows):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29500 + (os.getpid() * 7 + world * 13 + tile_rows) % 2000
    mp.spawn(_worker, args=(world, port, tile_rows, q), nprocs=world, join=True)
    assert q.get() is True


def test_frame_gather_single_rank(built):
    from rayz_amd import dist as rdist

    fg = rdist.FrameGather(10, 4, 1, 0, torch.device("cpu"))
    fg.tile.copy_(torch.arange(10 * 4 * 3, dtype=torch.float32).reshape(10, 4, 3))
    assert torch.equal(fg.gather(), fg.tile)
    assert rdist.DEFAULT_TILE_ROWS == 8  # = RAYZ_DEFAULT_TILE_ROWS, the library's one default (include/rayz_hip.h)
    assert rdist.max_shard_rows(1080, 8) == 136 and rdist.max_shard_rows(2160, 8) == 272
    assert rdist.max_shard_rows(1080, 8, tile_rows=1) == 135
