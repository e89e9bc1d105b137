"""The cross-rank steps of the data-parallel training step with 2 gloo ranks on the CPU: the one-time
parameter broadcast, the mean all-reduce of the flat gradient buffer (what Jittor's optimizer does under
MPI before the clip and the update, image_restoration_model.py:267-271) and the mean of the logged loss
(reduce_loss_dict, base_model.py:287-305).  After the exchange every rank must hold the same gradient, so
that identical updates keep the replicas identical."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jirf_hip.train import allreduce_flat_gradients, average_flat_gradients, broadcast_flat_parameters, reduce_loss


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _adam(p, g, m, v, step, lr=2e-4, b0=0.9, b1=0.999, eps=1e-8, clip=0.01):
    coef = min(clip / (float(np.sqrt(np.sum(g.astype(np.float64) ** 2))) + 1e-6), 1.0)
    g = g * np.float32(coef)
    m[:] = b0 * m + (1 - b0) * g
    v[:] = b1 * v + (1 - b1) * g * g
    return p - m * (lr * np.sqrt(1 - b1 ** step) / (1 - b0 ** step)) / (np.sqrt(v) + eps)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 100_003
        rng = np.random.default_rng(100 + rank)                      # every rank starts from other weights ...
        p = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
        broadcast_flat_parameters(p)                                 # ... until rank 0's are broadcast
        p0 = p.numpy().copy()
        m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
        for step in (1, 2, 3):
            g_local = np.random.default_rng(10 * step + rank).standard_normal(n).astype(np.float32) * 1e-3
            g = torch.from_numpy(g_local.copy())
            average_flat_gradients(g)
            p = torch.from_numpy(_adam(p.numpy(), g.numpy(), m, v, step).astype(np.float32))
        # what the training step does: the SUM stays in the buffer and 1 / ranks goes to the update as grad_scale
        gs = torch.from_numpy(g_local.copy())
        scale = allreduce_flat_gradients(gs)
        assert scale == 1.0 / world
        assert np.allclose(gs.numpy() * np.float32(scale), g.numpy(), rtol=0, atol=1e-9)
        loss = reduce_loss(1.0 + rank, "cpu")
        out.put((rank, p0, g.numpy().copy(), p.numpy().copy(), loss))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_average_keeps_replicas_identical():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0_a, g_a, p_a, l_a), (_, p0_b, g_b, p_b, l_b) = res
    assert np.array_equal(p0_a, p0_b)                                # broadcast: same starting weights
    want = 0.5 * (np.random.default_rng(30).standard_normal(100_003).astype(np.float32) * 1e-3 +
                  np.random.default_rng(31).standard_normal(100_003).astype(np.float32) * 1e-3)
    assert np.allclose(g_a, want, rtol=0, atol=1e-9)                 # the mean of the two local gradients
    assert np.array_equal(g_a, g_b)                                  # bit-identical on both ranks
    assert np.array_equal(p_a, p_b)                                  # so three clipped Adam steps keep them identical
    assert l_a == l_b == 1.5


def _solo_worker(port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = torch.arange(8, dtype=torch.float32)
        scale = allreduce_flat_gradients(g)                          # the collective runs over the one rank
        out.put((scale, g.numpy().copy(), reduce_loss(0.25, "cpu")))
    finally:
        dist.destroy_process_group()


def test_one_rank_group_still_takes_the_collective_path():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_solo_worker, args=(_free_port(), q))
    p.start()
    scale, g, loss = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert scale == 1.0 and np.array_equal(g, np.arange(8, dtype=np.float32)) and loss == 0.25


def test_single_process_is_a_no_op():
    g = torch.arange(8, dtype=torch.float32)
    average_flat_gradients(g)
    assert torch.equal(g, torch.arange(8, dtype=torch.float32))
    assert allreduce_flat_gradients(g) == 1.0
    assert reduce_loss(0.25, "cpu") == 0.25
