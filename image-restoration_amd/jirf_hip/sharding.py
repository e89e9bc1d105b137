"""Work partition for multi-GPU inference: whole images (or tiles) are independent units, so ranks
take disjoint index sets and no collective touches the data path.  The only cross-rank steps are
the barrier and the max-over-ranks of the elapsed time used for reporting."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Static round-robin: item i goes to rank i % world."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    return list(range(rank, n_items, world))


def shard_balanced(costs: Sequence[float], world: int) -> List[List[int]]:
    """For items of unequal cost (mixed image sizes): the longest-processing-time rule - items in order of falling
    cost (ties by index), each to the rank with the least work so far (ties by rank).  A function of `costs` alone,
    so every rank computes the same table without talking to the others; each rank's list is in rising index
    order.  With equal costs it is the round-robin of shard_indices."""
    if world < 1:
        raise ValueError(f"bad world {world}")
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda j: (-float(costs[j]), j)):
        r = min(range(world), key=lambda q: (load[q], q))
        out[r].append(i)
        load[r] += float(costs[i])
    return [sorted(o) for o in out]


def _group_state(group=None):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def gather_items(local: Dict[int, "object"], n_items: int, dst: int = 0, group=None) -> Optional[list]:
    """Collects per-item result tensors (float32, any shapes) from the ranks onto rank `dst`: returns the list
    [item 0, ..., item n_items - 1] there and None elsewhere.  One object all-gather of (index, shape) pairs and one
    tensor gather of each rank's results laid end to end (padded to the longest): two collectives per call, however
    many items.  Without a process group (or with one rank) it just orders the local results."""
    import torch
    import torch.distributed as dist
    rank, world = _group_state(group)
    if world == 1:
        if sorted(local) != list(range(n_items)):
            raise ValueError("gather_items: a single rank must hold every item")
        return [local[i] for i in range(n_items)]
    meta = [(int(i), tuple(int(d) for d in local[i].shape)) for i in sorted(local)]
    metas: List[Optional[list]] = [None] * world
    dist.all_gather_object(metas, meta, group=group)
    seen = sorted(i for m in metas for i, _ in m)
    if seen != list(range(n_items)):
        raise ValueError(f"gather_items: the ranks hold items {seen[:8]}... instead of 0..{n_items - 1} once each")

    def numel(shape):
        n = 1
        for d in shape:
            n *= d
        return n

    lengths = [sum(numel(sh) for _, sh in m) for m in metas]
    longest = max(max(lengths), 1)
    like = next(iter(local.values())) if local else None
    device = like.device if like is not None else (torch.device("cuda", torch.cuda.current_device())
                                                   if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    flat = torch.zeros(longest, dtype=torch.float32, device=device)
    off = 0
    for i, sh in meta:
        n = numel(sh)
        flat[off:off + n] = local[i].reshape(-1).to(torch.float32)
        off += n
    dst_global = dist.get_global_rank(group, dst) if group is not None else dst
    if rank != dst:
        dist.gather(flat, None, dst=dst_global, group=group)
        return None
    bufs = [torch.empty_like(flat) for _ in range(world)]
    dist.gather(flat, bufs, dst=dst_global, group=group)
    out: List[Optional[object]] = [None] * n_items
    for r, m in enumerate(metas):
        off = 0
        for i, sh in m:
            n = numel(sh)
            out[i] = bufs[r][off:off + n].reshape(sh).clone()
            off += n
    return out


def run_sharded(items: Sequence, fn: Callable[[List[int], list], list], costs: Optional[Sequence[float]] = None,
                gather: bool = True, dst: int = 0, group=None):
    """The multi-rank form of a test script's `for file in files` loop (Deraining/test_rain100L.py:141-190): every
    rank calls this with the same `items`; item i is processed by exactly one rank - i % ranks, or with `costs` the
    balanced table of shard_balanced - through fn(indices, [items[i] for i in indices]) -> list of result tensors,
    and the results are gathered onto rank `dst` in item order (gather=False: each rank keeps its own
    {index: result}).  No collective touches the data path; the gather is the only exchange."""
    rank, world = _group_state(group)
    mine = shard_indices(len(items), rank, world) if costs is None else shard_balanced(costs, world)[rank]
    results = fn(mine, [items[i] for i in mine]) if mine else []
    if len(results) != len(mine):
        raise ValueError(f"run_sharded: fn returned {len(results)} results for {len(mine)} items")
    local = dict(zip(mine, results))
    return gather_items(local, len(items), dst, group) if gather else local


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a scalar when a process group is up; identity otherwise."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
