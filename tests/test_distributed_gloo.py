"""CPU, world_size 2, gloo: the N>1 path (unit partition + result gather) of moshpp_amd.parallel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from moshpp_amd.parallel import run_sharded
    units = [dict(seq=i, frames=f) for i, f in enumerate([40, 10, 30, 20, 25, 5, 15])]

    def solve_local(us):   # stands in for StageIISolver.solve on this rank's GPU
        return [dict(seq=u['seq'], rank=rank, fullpose=np.full((u['frames'], 3), float(u['seq']))) for u in us]

    res = run_sharded(units, [u['frames'] for u in units], solve_local, dist=dist)
    if rank == 0:
        assert [r['seq'] for r in res] == list(range(7))
        assert all(r['fullpose'].shape == (u['frames'], 3) and (r['fullpose'] == u['seq']).all() for r, u in zip(res, units))
        ranks = sorted(set(r['rank'] for r in res))
        assert ranks == [0, 1]
        loads = [sum(u['frames'] for r, u in zip(res, units) if r['rank'] == k) for k in (0, 1)]
        assert abs(loads[0] - loads[1]) <= 10
        np.save(os.path.join(outdir, 'ok.npy'), np.array(loads))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok.npy')


# ---- one sequence sharded over ranks by frame ranges (solve_sequence_sharded) ----------------------------------------
def _toy_chain(obs, init, seed_state=None):
    """A stand-in for the Stage-II chain with the same structure: the state after frame t depends on the observation and on
    the two previous states (warm start + velocity term), contracts like the real chain (0.45 per frame), and a fresh start
    lands in a WRONG basin for frames 95..140 (it stays there until the basins merge at frame 141)."""
    F = len(obs)
    pose = np.zeros((F, 4)); trans = np.zeros((F, 3))
    if init is None:
        p1 = np.zeros(4); p2 = np.zeros(4); fresh = True
    else:
        p1, p2, fresh = np.array(init['pose']), np.array(init['pose_prev']), False
    wrong = False
    for t in range(F):
        g = obs[t]                                            # global frame number, doubles as the "observation"
        target = np.array([np.sin(0.05 * g), np.cos(0.03 * g), 0.01 * g, 1.0])
        if fresh and t == 0 and 95 <= g <= 140:
            wrong = True
        if g > 140:
            wrong = False
        basin = np.array([0.5, 0.0, 0.0, 0.0]) if wrong else 0.0
        if fresh and t == 0:
            p = target + basin + 0.03                         # first-frame schedule: close, not converged
        else:
            p = target + basin + 0.45 * ((p1 - (_tgt(g - 1) + (basin if not (fresh and t == 0) else 0))) ) + 0.0 * p2
        pose[t] = p; trans[t] = p[:3] * 0.1
        p2, p1 = p1, p
    return dict(pose=pose, trans=trans, status=np.zeros(F, dtype=np.int32))


def _tgt(g):
    return np.array([np.sin(0.05 * g), np.cos(0.03 * g), 0.01 * g, 1.0])


def _worker_frames(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from moshpp_amd.parallel import solve_sequence_sharded, frame_ranges
    F = 240
    calls = []

    def solve_range(a, b, init):
        calls.append((a, b, init is not None))
        return _toy_chain(np.arange(a, b), init)

    out, info = solve_sequence_sharded(solve_range, F, dist=dist, warmup=32, verify_tol=1e-9)
    a, b = frame_ranges(F, world)[rank]
    assert info['range'] == (a, b) and out['pose'].shape == (b - a, 4)
    ref = _toy_chain(np.arange(0, F), None)                   # the sequential chain over the whole sequence
    assert np.abs(out['pose'] - ref['pose'][a:b]).max() < 1e-8, np.abs(out['pose'] - ref['pose'][a:b]).max()
    gathered = [None] * world
    dist.all_gather_object(gathered, (calls, info))
    if rank == 0:
        # rank 1 owns [80, 160): its fresh start at frame 48 is fine, ... the start of rank 2 (frame 128) sits in the wrong
        # basin -> exactly that rank is repaired, from its neighbour's end state
        assert gathered[0][1]['repaired'] == [[2]], gathered[0][1]
        assert gathered[2][0] == [(128, 240, False), (160, 240, True)]
        assert gathered[1][0] == [(48, 160, False)]
        np.save(os.path.join(outdir, 'ok_frames.npy'), np.array([1]))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_gloo_one_sequence_by_frame_ranges(tmp_path):
    port = _free_port()
    mp.spawn(_worker_frames, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert os.path.exists(tmp_path / 'ok_frames.npy')


# ---- Stage-I: frames of one subject over ranks, normal equations all-reduced (moshii_stagei_desc.sharded) --------------------
def _stagei_worker(rank, world, port, outdir, solver='', on_device=None):
    """The Stage-I solver with its frames split over `world` ranks and gloo as the all-reduce.  On this CPU-only box the kernels run
    through the g++ emulation build of stagei.hip (tests/emu): same source, same host code, same sharding logic as the GPU library."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from moshpp_amd.parallel import stagei_solve_sharded
    from tests import helpers
    from tests.emu import emu_stagei
    if solver:
        os.environ['MOSHII_S1_SOLVER'] = solver
    c = helpers.stagei_case(M=24, F=5, n_verts=1500, seed=11)
    kw = helpers.stagei_kwargs(c)
    out = stagei_solve_sharded(lambda **sh: emu_stagei.solve(c['m'], c['prior'], **kw, **sh), len(c['frames']), dist, on_device=on_device, device_memory_is_host=True)
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), **{k: np.asarray(v) for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_stagei_frames_sharded_over_ranks_gloo(tmp_path):
    """world_size 2 and 3 (uneven split of 5 frames): every rank returns the same solution, equal to the single-process solve up to
    the summation order of the reduced normal equations."""
    sys.path.insert(0, ROOT)
    from tests import helpers
    from tests.emu import emu_stagei
    emu_stagei.build_emu.build()                 # compile once, before the ranks race for it
    c = helpers.stagei_case(M=24, F=5, n_verts=1500, seed=11)
    single = emu_stagei.solve(c['m'], c['prior'], **helpers.stagei_kwargs(c))
    for world in (2, 3):
        d = tmp_path / f'w{world}'
        os.makedirs(d)
        mp.spawn(_stagei_worker, args=(world, _free_port(), str(d), 'dense'), nprocs=world, join=True)
        outs = [np.load(d / f'rank{r}.npz') for r in range(world)]
        for o in outs:
            assert int(o['iters'][0]) == int(single['iters'][0])
            assert np.abs(o['betas'] - single['betas']).max() < 1e-9
            assert np.abs(o['markers_latent'] - single['markers_latent']).max() < 1e-10
            assert np.abs(o['pose'] - single['pose']).max() < 1e-9 and np.abs(o['trans'] - single['trans']).max() < 1e-10
            assert np.allclose(o['errs'], single['errs'], rtol=1e-8, atol=1e-12)
        for o in outs[1:]:
            assert np.array_equal(o['betas'], outs[0]['betas']) and np.array_equal(o['pose'], outs[0]['pose'])


def test_stagei_sharded_schur_allreduces_only_the_shared_block(tmp_path):
    """MOSHII_S1_SOLVER=schur with the frames over 2 and 3 ranks: every rank eliminates its own frames and only the Schur system of the
    shared block (ns^2 + ns doubles) plus a few n-vectors are summed -- the reduction SURVEY 8(e) describes.  Same solution as the
    single-process dense solve."""
    sys.path.insert(0, ROOT)
    from tests import helpers
    from tests.emu import emu_stagei
    emu_stagei.build_emu.build()
    c = helpers.stagei_case(M=24, F=5, n_verts=1500, seed=11)
    single = emu_stagei.solve(c['m'], c['prior'], **helpers.stagei_kwargs(c))
    for world in (2, 3):
        d = tmp_path / f'w{world}'
        os.makedirs(d)
        mp.spawn(_stagei_worker, args=(world, _free_port(), str(d), 'schur'), nprocs=world, join=True)
        outs = [np.load(d / f'rank{r}.npz') for r in range(world)]
        for o in outs:
            assert int(o['iters'][0]) == int(single['iters'][0])
            assert np.abs(o['betas'] - single['betas']).max() < 1e-9 and np.abs(o['markers_latent'] - single['markers_latent']).max() < 1e-10
            assert np.abs(o['pose'] - single['pose']).max() < 1e-9 and np.abs(o['trans'] - single['trans']).max() < 1e-10
        for o in outs[1:]:
            assert np.array_equal(o['betas'], outs[0]['betas']) and np.array_equal(o['pose'], outs[0]['pose'])


def _stagei_bad_owner_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from moshpp_amd.parallel import make_allreduce
    from tests import helpers
    from tests.emu import emu_stagei
    c = helpers.stagei_case(M=24, F=5, n_verts=1500, seed=11)
    kw = helpers.stagei_kwargs(c)
    rng = (0, 5) if rank == 0 else (5, 5)                       # rank 1 owns no frame, yet claims the shared rows
    try:
        emu_stagei.solve(c['m'], c['prior'], **kw, frame_range=rng, owns_shared_rows=(rank == 1), allreduce=make_allreduce(dist))
        msg = 'no error'
    except RuntimeError as e:
        msg = str(e)
    with open(os.path.join(outdir, f'rank{rank}.txt'), 'w') as f:
        f.write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_stagei_sharded_argument_errors_fail_on_every_rank_together(tmp_path):
    """A rank with bad sharding arguments does not return alone (the others would wait for it in the first all-reduce): the verdict is
    summed over the ranks and all of them fail with the same message."""
    sys.path.insert(0, ROOT)
    from tests.emu import emu_stagei
    emu_stagei.build_emu.build()
    mp.spawn(_stagei_bad_owner_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    msgs = [open(tmp_path / f'rank{r}.txt').read() for r in range(2)]
    assert all('owns the shared rows must own at least one frame' in m for m in msgs), msgs


@pytest.mark.parametrize('solver', ['schur', 'dense'])
def test_stagei_sharded_reduces_on_the_solvers_own_buffers(tmp_path, solver):
    """moshii_stagei_desc.allreduce_on_device: the callback is handed the solver's device buffers (the Schur block / the normal
    equations in place, vectors and scalars through a device scratch) -- what RCCL needs.  Here the "device" is the emulation's host
    memory and gloo sums it through the same pointers: the protocol, the buffer sizes and the result are what the GPU path uses."""
    sys.path.insert(0, ROOT)
    from tests import helpers
    from tests.emu import emu_stagei
    emu_stagei.build_emu.build()
    c = helpers.stagei_case(M=24, F=5, n_verts=1500, seed=11)
    single = emu_stagei.solve(c['m'], c['prior'], **helpers.stagei_kwargs(c))
    d = tmp_path / 'w2'
    os.makedirs(d)
    mp.spawn(_stagei_worker, args=(2, _free_port(), str(d), solver, True), nprocs=2, join=True)
    outs = [np.load(d / f'rank{r}.npz') for r in range(2)]
    for o in outs:
        assert int(o['iters'][0]) == int(single['iters'][0])
        assert np.abs(o['betas'] - single['betas']).max() < 1e-9 and np.abs(o['markers_latent'] - single['markers_latent']).max() < 1e-10
        assert np.abs(o['pose'] - single['pose']).max() < 1e-9 and np.abs(o['trans'] - single['trans']).max() < 1e-10
    assert np.array_equal(outs[0]['betas'], outs[1]['betas']) and np.array_equal(outs[0]['pose'], outs[1]['pose'])


# ---- bench.py's fixed (strong-scaling) job: which sequences a rank takes ---------------------------------------------------------
def _bench_share_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    mine = bench.strong_job_shares(32, 4000, world)[rank]          # every rank computes the partition for itself
    got = [None] * world
    dist.all_gather_object(got, mine)
    if rank == 0:
        np.save(os.path.join(outdir, 'shares.npy'), np.array(got, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_strong_job_is_partitioned_evenly_over_two_ranks(tmp_path):
    mp.spawn(_bench_share_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    shares = np.load(tmp_path / 'shares.npy', allow_pickle=True)
    flat = sorted(i for s in shares for i in s)
    assert flat == list(range(32))                                  # every sequence solved exactly once
    assert [len(s) for s in shares] == [16, 16]
    import bench
    assert [len(s) for s in bench.strong_job_shares(32, 4000, 8)] == [4] * 8
    assert bench.strong_job_shares(32, 4000, 1) == [list(range(32))]          # N = 1: the many_sequences leg
