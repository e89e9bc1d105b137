"""The multi-GPU path shards whole images across ranks with no data-path collective; the only
cross-rank steps are a barrier and a MAX/SUM of scalars.  Exercised here with 2 gloo ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jirf_hip.sharding import max_over_ranks, shard_indices, sum_over_ranks


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
