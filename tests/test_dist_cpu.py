"""The N>1 data-parallel path on CPU: world_size-2 gloo processes, contiguous shards, one all-gather."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from interactvlm_amd.dist import evaluate_sharded, gather_contacts, reduce_meters, shard_range

    lo, hi = shard_range(n_items, rank, world)
    # each "image" i yields a deterministic contact row; ranks own contiguous shards
    rows = [torch.full((6890,), float(i)) + torch.arange(6890) * 1e-4 for i in range(lo, hi)]
    local = torch.stack(rows) if rows else torch.zeros(0, 6890)
    allc = gather_contacts(local, n_items)
    meters = reduce_meters(torch.tensor([float(hi - lo), float(local.sum())]))
    # the bench's configs[2] job: shard, evaluate in chunks of 3, one gather
    calls = []

    def chunk(idx):
        calls.append(len(idx))
        return torch.stack([torch.full((6890,), float(i)) for i in idx]) if idx else torch.zeros(0, 6890)
    job = evaluate_sharded(n_items, 3, chunk)
    assert tuple(job.shape) == (n_items, 6890) and job[:, 0].tolist() == [float(i) for i in range(n_items)]
    assert all(c <= 3 for c in calls)
    if rank == 0:
        q.put((allc[:, 0].tolist(), allc.shape, meters.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items,world", [(4, 2), (8, 2), (5, 2), (1, 2), (10, 3), (2, 3)])
def test_gloo_shard_and_gather(n_items, world):
    """Divisible and NON-divisible item counts (uneven and even empty tail shards) through the real gather_contacts."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    first, shape, meters = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tuple(shape) == (n_items, 6890)
    assert first == [float(i) for i in range(n_items)]  # gathered in input order
    assert meters[0] == n_items


def test_single_process_is_identity():
    sys.path.insert(0, REPO)
    from interactvlm_amd.dist import gather_contacts, shard_range

    x = torch.randn(3, 6890)
    assert gather_contacts(x) is x
    assert shard_range(10, 3, 4) == (9, 10) and shard_range(10, 0, 4) == (0, 3) and shard_range(2, 3, 4) == (2, 2)


def _bench_worker(rank, world, port, n_img, q):
    """bench.py's N > 1 bookkeeping (timed region with the MAX over ranks, the one-GPU denominator of the dp64 job) on gloo with a
    stub chunk evaluator: rank r sleeps (r + 1) * 150 ms per chunk, so the slowest rank defines the time (sleeps long enough that
    the scheduling jitter of a busy 8-core container - tens of ms per call - stays well inside the tolerances)."""
    import time

    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from interactvlm_amd.dist import evaluate_sharded, shard_range

    calls = []

    def chunk(idx):
        calls.append(list(idx))
        time.sleep(0.15 * (rank + 1))
        return torch.stack([torch.full((6890,), float(i)) for i in idx]) if idx else torch.zeros(0, 6890)

    step = lambda: evaluate_sharded(n_img, 2, chunk)
    dt, res = bench.timed_steps(step, 1, 3, world, "cpu")
    assert tuple(res.shape) == (n_img, 6890) and res[:, 0].tolist() == [float(i) for i in range(n_img)]
    slowest = max(0.15 * (r + 1) * -(-(shard_range(n_img, r, world)[1] - shard_range(n_img, r, world)[0]) // 2) for r in range(world))
    assert dt >= 3 * slowest * 0.95, (dt, slowest)  # the MAX over ranks, not this rank's own time
    one = bench.one_gpu_same_workload(n_img, 2, chunk, rank, world, "cpu", res, dt, 3)
    if rank == 0:
        assert one is not None and one["max_abs_dp_sharded_vs_one_gpu"] == 0.0
        assert abs(one["images_per_s"] - n_img / one["seconds"]) < 2e-2 * one["images_per_s"]
        assert abs(one["speedup_of_this_run"] - (n_img * 3 / dt) / one["images_per_s"]) < 2e-2 * one["speedup_of_this_run"]
        # rank 0 alone runs every chunk at 150 ms; sharded, the slowest rank defines the step: the speedup is what the sleeps say
        expect = (0.15 * -(-n_img // 2)) / slowest
        assert abs(one["speedup_of_this_run"] - expect) < 0.3 * expect, (one, expect)
        q.put((dt, one))
    else:
        assert one is None
    dist.destroy_process_group()


@pytest.mark.parametrize("n_img,world", [(8, 2), (7, 3)])
def test_bench_multi_rank_bookkeeping_on_gloo(n_img, world):
    """the N > 1 branch of bench.py (all_reduce(MAX) of the time, the one_gpu_same_workload denominator and
    speedup_of_this_run) executed by gloo ranks with a stub chunk - the first 8-GPU box must not also debug the bookkeeping."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, n_img, q)) for r in range(world)]
    for p in procs:
        p.start()
    dt, one = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert dt > 0 and one["speedup_of_this_run"] > 0.3
