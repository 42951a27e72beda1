"""Multi-GPU sharding of a frame: interleaved row tiles, one process per GPU, one framebuffer gather.

The path shards embarrassingly (every pixel-sample has its own PCG32 stream keyed by the GLOBAL pixel index,
so the image does not depend on how rows are dealt out).  Rows go to ranks in interleaved tiles of `tile_rows`
rows (default 8, the library's own default: the BVH kernel's 8x8 work tiles are then 8 consecutive image rows — measured
+8.6 % / +1.7 % on an 8-way deal of the 1080p / 4K frame against 1-row interleave, profiles/r04/multi/tile_rows_ab.log;
contiguous bands would be badly balanced, sky rows finish in one segment per sample) and the only
collective is one all_gather of the ranks' compact row blocks at the end (RCCL over xGMI with backend "nccl";
the same code runs on gloo for the CPU tests).  The reference has no counterpart: it is single-threaded.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import capi, render

DEFAULT_TILE_ROWS = 8  # = RAYZ_DEFAULT_TILE_ROWS (include/rayz_hip.h): one default everywhere


def shard_params(params: capi.RenderParams, rank: int, world: int, tile_rows: int = DEFAULT_TILE_ROWS) -> capi.RenderParams:
    """A copy of `params` restricted to rank's rows."""
    p = capi.RenderParams.from_buffer_copy(bytes(params))
    p.tile_rows, p.shard_index, p.shard_count = tile_rows, rank, world
    return p


def max_shard_rows(height: int, world: int, tile_rows: int = DEFAULT_TILE_ROWS) -> int:
    return max(len(render.shard_row_indices(height, tile_rows, r, world)) for r in range(world))


class FrameGather:
    """Pre-allocated buffers + index maps for gathering row-tile shards into the full frame on every rank.

    `dtype=torch.uint8` applies `Image.writePPM`'s per-pixel transform (src/image.zig:35-38: sqrt-gamma, clamp,
    truncate x*255) to this rank's rows BEFORE the collective — `rayz_hip_tonemap_u8` on the GPU — so the tiles
    travel as u8, 4x smaller than f32, and the gathered frame is what writePPM prints.  The render still writes f32
    into `tile`; the u8 copy lives in `tile8`.  `tonemap` replaces the device transform (CPU tests)."""

    def __init__(self, height: int, width: int, world: int, rank: int, device, dtype=torch.float32,
                 tile_rows: int = DEFAULT_TILE_ROWS, group=None, tonemap=None):
        self.h, self.w, self.world, self.rank, self.group = height, width, world, rank, group
        self.rows = [render.shard_row_indices(height, tile_rows, r, world) for r in range(world)]
        self.max_rows = max(len(r) for r in self.rows)
        self.u8 = dtype == torch.uint8
        self._tonemap = tonemap
        # this rank's rows (padded) as the render writes them
        self.tile = torch.zeros((self.max_rows, width, 3), dtype=torch.float32 if self.u8 else dtype, device=device)
        self.tile8 = torch.zeros((self.max_rows, width, 3), dtype=torch.uint8, device=device) if self.u8 else None
        self.frame = torch.empty((height, width, 3), dtype=dtype, device=device)
        self._idx = [torch.as_tensor(r, device=device, dtype=torch.long) for r in self.rows]
        # concatenated along dim 0 (the form both RCCL and gloo accept); viewed per rank when scattering rows
        self._gathered = (torch.empty((world * self.max_rows, width, 3), dtype=dtype, device=device)
                          if world > 1 else None)

    @property
    def my_rows(self) -> np.ndarray:
        return self.rows[self.rank]

    def gather(self) -> torch.Tensor:
        """all_gather the tiles and un-interleave them; returns the full frame (valid on every rank)."""
        send = self.tile
        if self.u8:
            if self._tonemap is not None:
                self._tonemap(self.tile, self.tile8)
            else:  # on the stream the render was queued on (torch's current stream)
                render.tonemap_u8(self.tile.data_ptr(), self.tile8.data_ptr(), self.max_rows * self.w,
                                  torch.cuda.current_stream().cuda_stream)
            send = self.tile8
        if self.world == 1:
            self.frame.copy_(send[: self.h])
            return self.frame
        dist.all_gather_into_tensor(self._gathered, send, group=self.group)
        parts = self._gathered.view(self.world, self.max_rows, self.w, 3)
        for r in range(self.world):
            self.frame.index_copy_(0, self._idx[r], parts[r, : len(self.rows[r])])
        return self.frame


def frame_sha256(frame: torch.Tensor) -> str:
    """Fingerprint of a gathered frame: sha256 over its values, row-major RGB, in the frame's own dtype.  The image does not
    depend on the shard count, so bench.py's N = 1 and N > 1 lines must carry the same value (and the BVH frame the flat
    list's): a multi-GPU run proves itself."""
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(frame.detach().cpu().numpy()).tobytes()).hexdigest()


def rank_stats(values, device, world: int, group=None) -> np.ndarray:
    """Every rank's row of per-rank figures on every rank: one all_gather of a short f64 vector -> (world, len(values)).
    bench.py reports each rank's mean trace-kernel time, gather time and segment count with it, so that load imbalance
    between the row shards shows in the N > 1 line (the same code runs on gloo in the CPU tests)."""
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if world == 1:
        return mine.cpu().numpy()[None, :]
    out = torch.empty(world * mine.numel(), dtype=torch.float64, device=device)  # concatenated: the form RCCL AND gloo accept
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.view(world, mine.numel()).cpu().numpy()


def per_rank_block(per_rank: np.ndarray) -> dict:
    """bench.py's `per_rank` object from rank_stats' table (columns: kernel ms, gather ms, segments)."""
    k, g = per_rank[:, 0], per_rank[:, 1]
    return {
        "kernel_ms": {"min": float(k.min()), "mean": float(k.mean()), "max": float(k.max()), "all": [float(x) for x in k]},
        "gather_ms": {"min": float(g.min()), "mean": float(g.mean()), "max": float(g.max()),
                      "note": "a rank's own tile traced -> frame assembled on that rank: the all_gather (transfer + waiting for the "
                              "slowest rank) + the un-interleave; N = 1: a device copy"},
        "segments": [float(x) for x in per_rank[:, 2]],
    }
