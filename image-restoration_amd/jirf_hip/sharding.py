"""Work partition for multi-GPU inference: whole images (or tiles) are independent units, so ranks
take disjoint index sets and no collective touches the data path.  The only cross-rank steps are
the barrier and the max-over-ranks of the elapsed time used for reporting."""
from __future__ import annotations

from typing import List


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Static round-robin: item i goes to rank i % world."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    return list(range(rank, n_items, world))


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
