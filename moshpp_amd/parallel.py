"""Multi-GPU sharding of Stage-II work: one process per GPU, no data-path collective.

The reference fans out one OS process per capture (src/moshpp/mosh_head.py:584-589, run_tools.py:45-67).
Here a *unit* is a chain (a whole sequence, or one chunk of a sequence in chunked mode); units are
independent, so ranks take a balanced contiguous slice of the unit list, solve it on their own GPU and the
per-unit results are gathered on rank 0 (`torch.distributed` object gather: RCCL on GPUs, gloo on CPU tests).
"""
from __future__ import annotations

from typing import Callable, List, Sequence


def partition_units(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time greedy partition of unit indices by cost (e.g. frames per chain).
    Deterministic; every rank computes the same answer.  Returns world_size lists of unit indices,
    each sorted ascending."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world_size
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += float(costs[i])
    return [sorted(p) for p in parts]


def run_sharded(units: Sequence, costs: Sequence[float], solve_local: Callable[[List], List], dist=None):
    """Solve `units` across the ranks of the initialised process group (or locally when dist is None).
    `solve_local(list_of_units) -> list_of_results` runs on this rank's GPU.
    Returns the full, unit-ordered result list on rank 0 and None elsewhere."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return solve_local(list(units))
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = partition_units(costs, world)
    mine = parts[rank]
    local = solve_local([units[i] for i in mine])
    assert len(local) == len(mine)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(list(zip(mine, local)), gathered, dst=0)
    if rank != 0:
        return None
    out = [None] * len(units)
    for chunk in gathered:
        for i, res in chunk:
            out[i] = res
    assert all(o is not None for o in out)
    return out
