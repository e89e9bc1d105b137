"""The multi-GPU path shards whole images across ranks with no data-path collective; the only
cross-rank steps are a barrier and a MAX/SUM of scalars.  Exercised here with 2 gloo ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jirf_hip.sharding import gather_items, max_over_ranks, run_sharded, shard_balanced, shard_indices, sum_over_ranks


def test_shard_indices_partition():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 4, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_indices(11, rank, world)
        t = max_over_ranks(0.5 + rank)
        n = sum_over_ranks(float(len(mine)))
        dist.barrier()
        out.put((rank, mine, t, n))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res[0][1] + res[1][1]) == list(range(11))
    assert not set(res[0][1]) & set(res[1][1])
    assert res[0][2] == res[1][2] == 1.5            # MAX over ranks
    assert res[0][3] == res[1][3] == 11.0           # every item counted once


def test_shard_balanced_is_a_partition_and_evens_out_mixed_sizes():
    costs = [481 * 321] * 5 + [1280 * 720] * 3 + [256 * 256] * 9
    for world in (1, 2, 4, 8):
        parts = shard_balanced(costs, world)
        assert sorted(i for p in parts for i in p) == list(range(len(costs)))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)                  # the longest-processing-time bound
        assert all(p == sorted(p) for p in parts)
    # equal costs: the plain round-robin
    assert shard_balanced([1.0] * 10, 4) == [shard_indices(10, r, 4) for r in range(4)]
    # round-robin would give rank 0 both large images
    assert max(sum(c for c in [9, 1, 9, 1][r::2]) for r in range(2)) == 18
    assert [sum([9, 1, 9, 1][i] for i in p) for p in shard_balanced([9, 1, 9, 1], 2)] == [10, 10]


def _item(i):
    """Item i of a folder of mixed sizes: a [3, h, w] image whose content names its index."""
    h, w = 4 + i % 3, 5 + (i * 7) % 4
    return torch.arange(3 * h * w, dtype=torch.float32).reshape(3, h, w) + 1000.0 * i


def _restore(indices, images):
    return [2.0 * x[:, :-1, :-1] + 1.0 for x in images]               # changes the shape, like pad -> crop does not


def _sharded_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = [_item(i) for i in range(9)]
        calls = []

        def fn(indices, images):
            calls.append(list(indices))
            return _restore(indices, images)

        got = run_sharded(items, fn)                                  # round-robin, gathered on rank 0
        costs = [float(x.shape[1] * x.shape[2]) for x in items]
        got_b = run_sharded(items, fn, costs=costs, dst=1)            # balanced by pixel count, gathered on rank 1
        kept = run_sharded(items, fn, gather=False)
        empty = run_sharded(items[:1], fn)                            # fewer items than ranks: rank 1 holds none
        out.put((rank, calls, None if got is None else [g.numpy() for g in got],
                 None if got_b is None else [g.numpy() for g in got_b], sorted(kept),
                 None if empty is None else [e.numpy() for e in empty]))
    finally:
        dist.destroy_process_group()


def test_two_rank_run_sharded_gathers_every_item_in_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [w.numpy() for w in _restore(None, [_item(i) for i in range(9)])]
    (_, calls0, got0, gotb0, kept0, empty0), (_, calls1, got1, gotb1, kept1, empty1) = res
    assert got1 is None and gotb0 is None and empty1 is None
    for got in (got0, gotb1):
        assert len(got) == 9 and all(g.shape == w.shape and (g == w).all() for g, w in zip(got, want))
    assert calls0[0] == [0, 2, 4, 6, 8] and calls1[0] == [1, 3, 5, 7]
    assert sorted(calls0[1] + calls1[1]) == list(range(9))            # the balanced table is a partition too
    assert kept0 == [0, 2, 4, 6, 8] and kept1 == [1, 3, 5, 7]
    assert len(empty0) == 1 and (empty0[0] == want[0]).all()
    assert calls1[3:] == []                                           # rank 1 had nothing to do for one item


def test_run_sharded_without_a_process_group_is_the_plain_loop():
    items = [_item(i) for i in range(4)]
    got = run_sharded(items, _restore)
    assert all((g == w).all() for g, w in zip(got, _restore(None, items)))
    with pytest.raises(ValueError):
        gather_items({0: items[0]}, 2)
