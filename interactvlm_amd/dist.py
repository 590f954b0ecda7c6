"""Data-parallel glue: the path shards over images (one process per GPU, weights + lift tables replicated) and
needs exactly one collective per batch — the all-gather of per-vertex contacts (reference: evaluate.py:202-222,
torch.distributed.all_gather of [N_local, 6890] predictions).  On MI355X this is RCCL over xGMI; the message is
27.5 KB per image, i.e. latency-bound, so a single fused all_gather_into_tensor is used, never a per-sample loop.
Shards may be uneven: the collective always moves ceil(n / world) rows per rank (padding on the tail ranks).
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


def gather_contacts(local: torch.Tensor, n_items: int | None = None, group=None) -> torch.Tensor:
    """local [B_local, Nv] fp32 (this rank's contiguous shard, ``shard_range`` order) -> [n_items, Nv] on every rank with
    ONE all-gather.  Shards may be uneven (n_items % world != 0, even empty on the last ranks): every rank pads its block
    to ceil(n_items / world) rows - the collective then has equal contributions - and the padding is trimmed afterwards.
    n_items None: every rank holds the same number of rows (world * B_local results)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if n_items is None:
        per = local.shape[0]
        n_items = per * world
    else:
        per = (n_items + world - 1) // world
        lo, hi = shard_range(n_items, rank, world)
        if local.shape[0] != hi - lo:
            raise ValueError(f"rank {rank}: {local.shape[0]} rows, its shard of {n_items} items has {hi - lo}")
    send = local.contiguous()
    if send.shape[0] != per:  # pad the short (or empty) tail shard
        send = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]].copy_(local)
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send, group=group)
    return out[:n_items]  # contiguous shards + tail padding: the real rows are exactly the first n_items


def evaluate_sharded(n_items: int, per_call: int, eval_chunk, rank: int | None = None, world: int | None = None, group=None):
    """The data-parallel job of BASELINE.json configs[2] (reference: evaluate.py:202-210, 346): this rank's contiguous shard
    of the items [0, n_items) is evaluated ``per_call`` items at a time by ``eval_chunk(indices) -> [len(indices), Nv]`` (or a
    function returning that: a deferred chunk, see below), then ONE all-gather brings every rank the [n_items, Nv] result in input order."""
    ini = dist.is_available() and dist.is_initialized()
    rank = (dist.get_rank(group) if ini else 0) if rank is None else rank
    world = (dist.get_world_size(group) if ini else 1) if world is None else world
    lo, hi = shard_range(n_items, rank, world)
    # eval_chunk may return its [len(indices), Nv] result or a FUNCTION that returns it (a deferred chunk: its independent work is
    # already enqueued - InteractVLMForCausalLM.evaluate_batch(deferred=True)): chunk c + 1 is begun before chunk c is finished,
    # so that the next chunk's SAM encoder runs under the previous chunk's mask-decoder launches
    parts, pending = [], None
    for i in range(lo, hi, per_call):
        r = eval_chunk(list(range(i, min(i + per_call, hi))))
        if pending is not None:
            parts.append(pending() if callable(pending) else pending)
        pending = r
    if pending is not None:
        parts.append(pending() if callable(pending) else pending)
    if parts:
        local = torch.cat(parts, 0)
    else:  # an empty tail shard still takes part in the collective
        probe = eval_chunk([])
        local = probe if probe is not None else torch.zeros(0)
    if world == 1:  # (also when a multi-rank job asks one rank for the whole job: no collective is entered)
        return local
    return gather_contacts(local, n_items, group=group)


def reduce_meters(values: torch.Tensor, group=None) -> torch.Tensor:
    """AverageMeter.all_reduce equivalent (utils/utils.py:176-198): SUM of a small fp32 vector."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
    return values
