"""Row sharding across ranks (one process per GPU) and the final gather.

Every surface is independent, so the [N,5] parameter rows (or the (z, c) rows) are split into
contiguous blocks, one per rank; grid constants and decoder weights are replicated.  The only
exchange is one all-gather of the implied-vol tensor and the flag words at the end (RCCL over
xGMI with backend "nccl"; "gloo" on CPU for tests).  Ranks may hold different row counts, so the
gather pads to the largest shard and trims afterwards.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous [lo, hi) block of rows for ``rank``: n // world each, remainder to the last."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base = n // world
    lo = rank * base
    hi = n if rank == world - 1 else lo + base
    return lo, hi


def shard_rows(x: torch.Tensor, world: int | None = None, rank: int | None = None) -> torch.Tensor:
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(x.shape[0], world, rank)
    return x[lo:hi]


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather row blocks produced from ``shard_bounds`` back into one [n_total, ...] tensor."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_total, world, r) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    pad_shape = (longest,) + tuple(local.shape[1:])
    buf = local.new_zeros(pad_shape)
    buf[: local.shape[0]] = local
    out = local.new_empty((world * longest,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    parts = [out[r * longest: r * longest + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)
