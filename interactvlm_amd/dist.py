"""Data-parallel glue: the path shards over images (one process per GPU, weights + lift tables replicated) and
needs exactly one collective per batch — the all-gather of per-vertex contacts (reference: evaluate.py:202-222,
torch.distributed.all_gather of [N_local, 6890] predictions).  On MI355X this is RCCL over xGMI; the message is
27.5 KB per image, i.e. latency-bound, so a single fused all_gather_into_tensor is used, never a per-sample loop.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n_items for this rank (DistributedSampler(shuffle=False) order is strided in
    the reference, evaluate.py:346; contiguous shards keep the gathered tensor in input order)."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def gather_contacts(local: torch.Tensor, group=None) -> torch.Tensor:
    """local [B_local, Nv] fp32 -> [world * B_local, Nv] on every rank (single all-gather)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    local = local.contiguous()
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out


def reduce_meters(values: torch.Tensor, group=None) -> torch.Tensor:
    """AverageMeter.all_reduce equivalent (utils/utils.py:176-198): SUM of a small fp32 vector."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
    return values
