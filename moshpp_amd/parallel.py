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


def frame_ranges(F: int, world_size: int) -> List[tuple]:
    """Contiguous frame ranges [a_r, b_r) of one sequence, one per rank (sizes differ by at most one frame)."""
    base, extra = divmod(int(F), int(world_size))
    out, a = [], 0
    for r in range(world_size):
        b = a + base + (1 if r < extra else 0)
        out.append((a, b))
        a = b
    return out


def solve_sequence_sharded(solve_range: Callable, F: int, dist=None, warmup: int = 32, verify_tol: float = 1e-9,
                           max_rounds: int = None):
    """ONE long sequence over the ranks of the process group: the frames of a sequence partitioned across the GPUs of a
    node (BASELINE configs 2 / 5), stitched so that the result equals the sequential chain's to `verify_tol`.

    It is the chunk scheme of moshii_sequence_solve one level up.  Rank r owns frames [a_r, b_r).
      pass 1   rank r > 0 starts `warmup` frames early with the first-frame schedule (chmosh.py:629-655) and discards those
               frames; rank 0 starts at frame 0 as the reference does.
      verify   the chain state after frame t is (pose_t, pose_{t-1}, trans_t).  Rank r's warm-up rows at a_r - 1, a_r - 2 are
               compared with rank r-1's rows for the same frames (one small all_gather of 2 NP + 3 doubles per rank).
      repair   a rank whose left hand-off misses re-solves its range from rank r-1's end state (warm start + velocity term
               exactly as :624-626, 656-657); repeated until every hand-off verifies (at most world_size - 1 rounds).
    No collective on the data path: only the boundary rows travel.

    solve_range(a, b, init) -> dict(pose[b-a, NP], trans[b-a, 3], ...per-frame arrays...) solves frames [a, b) on this
    rank's GPU; init = None (first-frame schedule) or dict(pose, trans, pose_prev) to continue a chain.
    Returns (out, info): `out` holds this rank's frames [a_r, b_r) (arrays sliced to the owned range), info =
    dict(range=(a_r, b_r), rounds, repaired=[ranks re-solved per round], max_handoff_dev)."""
    import numpy as np
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = solve_range(0, F, None)
        return out, dict(range=(0, F), rounds=0, repaired=[], max_handoff_dev=0.0)
    world, rank = dist.get_world_size(), dist.get_rank()
    ranges = frame_ranges(F, world)
    a, b = ranges[rank]
    if min(e - s for s, e in ranges) < 2:
        raise ValueError('solve_sequence_sharded needs at least two frames per rank')
    lead = min(warmup, a) if rank > 0 else 0
    full = solve_range(a - lead, b, None)                       # pass 1 (incl. the warm-up frames)

    def cut(res, skip):
        return {k: (v[skip:] if hasattr(v, 'shape') and len(v) == (b - a + skip) else v) for k, v in res.items()}

    def edge(res, upto):
        """Chain state after the last solved frame below local index `upto`: (pose, previous solved pose, trans), or None
        when fewer than two frames were solved there (frames without markers leave their rows untouched, :586-588)."""
        st = res.get('status')
        ok = np.arange(upto) if st is None else np.flatnonzero(np.asarray(st[:upto]) != 1)
        if len(ok) < 2:
            return None
        i, j = int(ok[-1]), int(ok[-2])
        return np.concatenate([res['pose'][i], res['pose'][j], res['trans'][i]])

    # the state with which this rank ENTERED its first owned frame (from its own warm-up), and the state it ends in
    entry = edge(full, lead) if rank > 0 else None
    out = cut(full, lead)
    rounds, repaired, max_dev = 0, [], 0.0
    max_rounds = world if max_rounds is None else max_rounds
    while True:
        final = edge(out, b - a)
        finals = [None] * world
        dist.all_gather_object(finals, final)
        miss = False
        # failures are decided from data every rank holds (`finals`, the gathered flags), so all ranks raise together: one rank
        # raising alone would leave the others waiting in the next collective
        starved = [r for r in range(world - 1) if finals[r] is None]
        if starved:
            raise RuntimeError('rank(s) %s solved fewer than two frames: no state to hand over' % starved)
        if rank > 0:
            left = finals[rank - 1]
            dev = float('inf') if entry is None else float(np.max(np.abs(entry - left)))
            miss = not (dev <= verify_tol)
            if not miss:
                max_dev = max(max_dev, dev)
        flags = [None] * world
        dist.all_gather_object(flags, bool(miss))
        if not any(flags):
            break
        if rounds >= max_rounds:    # (`flags` and `rounds` are the same on every rank)
            raise RuntimeError('sharded sequence solve did not converge: hand-offs still failing after %d rounds' % rounds)
        repaired.append([r for r, f in enumerate(flags) if f])
        if miss:   # re-solve the owned range from the left neighbour's end state
            NP = out['pose'].shape[1]
            init = dict(pose=left[:NP], pose_prev=left[NP:2 * NP], trans=left[2 * NP:])
            out = solve_range(a, b, init)
            entry = left.copy()
        rounds += 1
    devs = [None] * world
    dist.all_gather_object(devs, max_dev)
    return out, dict(range=(a, b), rounds=rounds, repaired=repaired, max_handoff_dev=float(max(devs)))


# ---- Stage-I: frames of one subject over the ranks ----------------------------------------------------------------
def make_allreduce(dist, device=None):
    """In-place sum of a 1-D float64 NumPy array over the ranks of `dist` (torch.distributed).  gloo reduces the host buffer
    directly; nccl (= RCCL over xGMI) goes through a device tensor."""
    import torch

    def allreduce(arr):
        t = torch.from_numpy(arr)
        if dist.get_backend() == 'nccl':
            g = t.to(device if device is not None else 'cuda')
            dist.all_reduce(g)
            t.copy_(g.cpu())
        else:
            dist.all_reduce(t)
    return allreduce


def make_allreduce_device(dist, device_memory_is_host=False):
    """In-place sum of `count` float64 values at a DEVICE address over the ranks: torch.distributed (backend nccl = RCCL over xGMI)
    on a tensor wrapped around the pointer -- nothing is copied, nothing visits the host (moshii_stagei_desc.allreduce_on_device).
    With another backend the call is refused unless `device_memory_is_host` says the "device" pointers are host memory (the CPU
    emulation of the kernels, tests only): the address is then wrapped as a NumPy view."""
    import numpy as np
    import torch

    class _DevArray:                      # the CUDA array interface is how torch adopts foreign device memory (HIP included)
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {'shape': (count,), 'typestr': '<f8', 'data': (ptr, False), 'version': 2}

    def allreduce(ptr, count):
        if dist.get_backend() == 'nccl':
            t = torch.as_tensor(_DevArray(ptr, count), device='cuda')
            dist.all_reduce(t)
            torch.cuda.current_stream().synchronize()
        else:
            import ctypes
            if not device_memory_is_host:   # a real GPU build: the pointer is device memory, gloo would read it on the host
                raise RuntimeError('make_allreduce_device needs the nccl (RCCL) backend on a GPU build (device_memory_is_host=True only '
                                   'for the CPU emulation of the kernels); use make_allreduce for ' + str(dist.get_backend()))
            arr = np.ctypeslib.as_array((ctypes.c_double * count).from_address(ptr))
            dist.all_reduce(torch.from_numpy(arr))
    return allreduce


def stagei_solve_sharded(solve, n_frames, dist, on_device=None, device_memory_is_host=False):
    """One Stage-I problem over the ranks: rank r evaluates frames frame_ranges(n_frames, world)[r], rank 0 the shared rows; the
    normal equations are summed with an all-reduce per dogleg iteration (moshii_stagei_desc.sharded) and every rank returns the
    same solution.  `solve(frame_range=..., owns_shared_rows=..., allreduce=...)` is capi.stagei_solve_host with the problem bound.
    on_device (default: with the nccl backend): the reduction runs on the solver's device buffers (make_allreduce_device)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = frame_ranges(n_frames, world)[rank]
    if on_device is None:                 # RCCL works on device memory: hand it the solver's buffers
        on_device = dist.get_backend() == 'nccl'
    if on_device:
        return solve(frame_range=(lo, hi), owns_shared_rows=(rank == 0), allreduce=make_allreduce_device(dist, device_memory_is_host), allreduce_on_device=True)
    return solve(frame_range=(lo, hi), owns_shared_rows=(rank == 0), allreduce=make_allreduce(dist))


# ---- one sequence as concurrent chunks, stitched on the host (for chain variants moshii_sequence_solve does not cover) --------
def solve_sequence_chunked_host(solve_ranges: Callable, F: int, n_chunks: int, warmup: int = 32, verify_tol: float = 1e-9,
                                max_rounds: int = None, state_keys=('pose', 'trans')):
    """The chunk scheme of moshii_sequence_solve driven from the host over a batched chain solve -- for the extended chain
    variant (free expression / DMPL coefficients), whose extra state moshii_sequence_solve does not hand over.

    solve_ranges([(a, b, init), ...]) -> [out, ...] solves all listed frame ranges concurrently (one chain each, one
    moshii_chain_solve call); init = None (first-frame schedule) or the dict `edge_state` returns.  Every out holds per-frame
    arrays incl. 'pose', 'trans', 'status' and whatever `state_keys` names (e.g. 'shape').
      pass 1   chunk c > 0 starts `warmup` frames early with the first-frame schedule; the warm-up rows are discarded.
      verify   the state with which chunk c entered its first frame (from its own warm-up) against the end state of chunk c-1.
      repair   every chunk whose hand-off misses is re-solved from its left neighbour's end state; all such chunks of a round run
               concurrently; repeated until every hand-off verifies (a fix-point: at most n_chunks - 1 rounds).
    Returns (out over frames [0, F), info = dict(n_chunks, rounds, repaired=[chunks per round], max_handoff_dev))."""
    import numpy as np
    C = max(1, min(int(n_chunks), F // 2 if F >= 4 else 1))
    ranges = frame_ranges(F, C)

    def edge_state(res, upto):
        st = res.get('status')
        ok = np.arange(upto) if st is None else np.flatnonzero(np.asarray(st[:upto]) != 1)
        if len(ok) < 2:
            return None
        i, j = int(ok[-1]), int(ok[-2])
        s = dict(pose=np.array(res['pose'][i]), pose_prev=np.array(res['pose'][j]), trans=np.array(res['trans'][i]))
        for k in state_keys:
            if k not in ('pose', 'trans'):
                s[k] = np.array(res[k][i])
        return s

    def deviation(x, y):
        return max(float(np.max(np.abs(np.asarray(x[k]) - np.asarray(y[k])))) if np.size(x[k]) else 0.0 for k in x)

    leads = [0] + [min(warmup, a) for a, _ in ranges[1:]]
    first = solve_ranges([(a - l, b, None) for (a, b), l in zip(ranges, leads)])
    entries = [None] + [edge_state(r, l) for r, l in zip(first[1:], leads[1:])]
    outs = [{k: (v[l:] if hasattr(v, 'shape') and len(v) == (b - a + l) else v) for k, v in r.items()}
            for r, l, (a, b) in zip(first, leads, ranges)]
    rounds, repaired, max_dev = 0, [], 0.0
    max_rounds = C if max_rounds is None else max_rounds
    while True:
        finals = [edge_state(o, b - a) for o, (a, b) in zip(outs, ranges)]
        todo = []
        for c in range(1, C):
            if finals[c - 1] is None:
                raise RuntimeError(f'chunk {c - 1} solved fewer than two frames: no state to hand over')
            dev = float('inf') if entries[c] is None else deviation(entries[c], finals[c - 1])
            if dev <= verify_tol:
                max_dev = max(max_dev, dev)
            else:
                todo.append(c)
        if not todo:
            break
        if rounds >= max_rounds:
            raise RuntimeError(f'chunked solve did not converge: hand-offs still failing after {rounds} rounds')
        redo = solve_ranges([(ranges[c][0], ranges[c][1], finals[c - 1]) for c in todo])
        for c, r in zip(todo, redo):
            outs[c] = r
            entries[c] = {k: np.array(v) for k, v in finals[c - 1].items()}
        repaired.append(list(todo))
        rounds += 1
    keys = [k for k, v in outs[0].items() if hasattr(v, 'shape') and len(v) == ranges[0][1] - ranges[0][0]]
    out = {k: np.concatenate([o[k] for o in outs], axis=0) for k in keys}
    return out, dict(n_chunks=C, rounds=rounds, repaired=repaired, max_handoff_dev=max_dev)
