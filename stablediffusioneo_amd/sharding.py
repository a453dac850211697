"""Multi-GPU layout of the canny2image path: one process per GPU, whole images per rank (image i -> rank
i % world_size), a full weight replica per rank, no collective inside the DDIM loop; the only exchange is one
all-gather of the final latents (RCCL over xGMI when the backend is "nccl"; 32 KiB per 512x512 image, so it is
latency-bound and the ring/bucket sizing of larger collectives does not apply).  The reference has no
distributed code at all; this is new functionality required by BASELINE.json configs[2]."""
from __future__ import annotations

from typing import List

import torch


def shard_indices(total: int, rank: int, world: int) -> List[int]:
    """Indices of the images this rank generates: rank, rank+world, ... (< total)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, total, world))


def unit_index(rank: int, step: int, world: int) -> int:
    """Index of the work unit (one image, or one batch of images) rank `rank` processes at its `step`-th step: the inverse of
    shard_indices -- shard_indices(total, rank, world)[step] == unit_index(rank, step, world).  bench.py seeds its inputs with it."""
    return rank + step * world


def gather_latents(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """local: (n_local, C, h, w) latents of shard_indices(total, rank, world) in that order.
    Returns (total, C, h, w) in image-index order on every rank.  Ragged shards (total % world != 0) are padded
    to the longest shard for the collective and trimmed afterwards."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        assert local.shape[0] == total
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (total + world - 1) // world
    n_local = len(shard_indices(total, rank, world))
    assert local.shape[0] == n_local, (local.shape, n_local)
    if n_local < per:
        pad = torch.zeros((per - n_local, *local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad])
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local.contiguous(), group=group)
    out = torch.empty((total, *local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(total, r, world)
        out[idx] = parts[r][: len(idx)]
    return out
