"""Multi-GPU sharding of a VRF batch: independent items, one process per GPU.

Items are independent, so the data path has no collective: rank g owns
the contiguous slice [g*N/G, (g+1)*N/G).  The only exchange is the result gather (status bytes /
proof bytes) over torch.distributed (RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice of rank `rank`; slices tile [0, n) exactly, sizes differ by <= 1."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    lo = n_items * rank // world
    hi = n_items * (rank + 1) // world
    return lo, hi


def gather_results(local, n_items: int, rank: int, world: int):
    """All-gather per-item results (tensor [n_local, ...]) into item order on every rank.

    Ragged slice sizes are handled by padding to the largest slice; one collective call."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local
    sizes = [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    parts = [out[r * m: r * m + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)
