"""Multi-GPU sharding of one frame: one process per GPU, scene replicated, no
communication on the data path, one collective on the radiance sum at the end.

The reference parallelises the same loop over CPU threads by contiguous column
chunks (RayTracer::traceImage, RayTracer.cpp:482-487); pixels and samples are
independent, so the frame splits either way:

  * "samples": every rank renders all pixels for its own slice of the sample range;
    the per-pixel sums are added with one reduce (RCCL `reduce(SUM)` on GPU tensors).
  * "tiles": every rank renders all samples of its own bands of rows (band b belongs
    to rank b % world); the disjoint bands are gathered to the root.

Per-sample clamping happens on the rank that owns the sample, before any sum, so both
modes give the single-GPU image (tiles: bit for bit; samples: up to fp64 summation
order).  Works on any torch.distributed backend: "nccl" (= RCCL) on GPUs, "gloo" in
the CPU tests.
"""
from __future__ import annotations

import copy
from typing import List

import numpy as np

from . import abi


def shard_params(base: abi.Params, rank: int, world: int, mode: str = "samples") -> abi.Params:
    """Render parameters of `rank` out of `world` for the frame `base` describes."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    p = copy.copy(base)
    if mode == "samples":
        n = base.sample_end - base.sample_begin
        p.sample_begin = base.sample_begin + (n * rank) // world
        p.sample_end = base.sample_begin + (n * (rank + 1)) // world
        p.shard_index, p.shard_count = base.shard_index, base.shard_count
    elif mode == "tiles":
        if base.shard_count != 1:
            raise ValueError("tiles mode shards an unsharded frame")
        p.shard_index, p.shard_count = rank, world
    else:
        raise ValueError("mode must be 'samples' or 'tiles'")
    return p


def owned_rows(p: abi.Params) -> np.ndarray:
    """Rows y with (y / band_rows) % shard_count == shard_index (include/mipt.h)."""
    y = np.arange(p.height)
    return y[(y // p.band_rows) % p.shard_count == p.shard_index]


def combine(d_sum, base: abi.Params, mode: str, dst: int = 0):
    """Bring the per-rank radiance sums (a float64 torch tensor of 3*w*h elements on each
    rank, zero outside the rank's shard) together on rank `dst`.  Returns the full-frame
    tensor on `dst`, None elsewhere.  The only collective of the render."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return d_sum
    world, rank = dist.get_world_size(), dist.get_rank()
    if mode == "samples":
        dist.reduce(d_sum, dst=dst, op=dist.ReduceOp.SUM)
        return d_sum if rank == dst else None
    if mode != "tiles":
        raise ValueError("mode must be 'samples' or 'tiles'")
    w, h = base.width, base.height
    frame = d_sum.view(h, w * 3)
    rows: List[np.ndarray] = [owned_rows(shard_params(base, r, world, "tiles")) for r in range(world)]
    most = max(len(r) for r in rows)
    mine = torch.zeros((most, w * 3), dtype=frame.dtype, device=frame.device)
    idx = torch.as_tensor(rows[rank], dtype=torch.long, device=frame.device)
    if len(rows[rank]):
        mine[: len(rows[rank])] = frame.index_select(0, idx)
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, gather_list=bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.zeros_like(frame)
    for r in range(world):
        if len(rows[r]):
            ridx = torch.as_tensor(rows[r], dtype=torch.long, device=frame.device)
            out.index_copy_(0, ridx, bufs[r][: len(rows[r])])
    return out.view(-1)
