"""Ensemble sharding across GPUs: one process per GPU (torchrun), contiguous blocks of the
trajectory index per rank, no exchange while stepping (trajectories are independent), one
all-gather of the per-shard results at the end (`torch.distributed` backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).  SURVEY.md 8(e)."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; the first `total % world` ranks get one extra.  The arithmetic is the
    C ABI's (`odef_shard_range`, include/odefilter.h), so the one-process-per-GPU path here and the single-process
    `odef_group` a Julia host drives cut an ensemble identically."""
    from .host import shard_range

    lo, n = shard_range(total, world, rank)
    return lo, lo + n


def init_from_env(backend: str = "nccl") -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allgather_shards(local: torch.Tensor, world: int) -> torch.Tensor:
    """The single collective of the path: gathers equally sized per-rank result blocks
    [..., n_local] into [world, ..., n_local]."""
    if world == 1:
        return local.unsqueeze(0)
    flat = local.contiguous().view(-1)
    out = torch.empty(world * flat.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, flat)  # 1-D concat form: accepted by both RCCL and gloo
    return out.view((world,) + tuple(local.shape))


def gathered_to_global(g: torch.Tensor) -> torch.Tensor:
    """[world, K, n_local] -> [K, world*n_local] in global trajectory order."""
    w, k, n = g.shape
    return g.permute(1, 0, 2).reshape(k, w * n)
