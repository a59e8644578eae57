"""One sequence sharded over ranks by frame ranges (moshpp_amd.parallel.solve_sequence_sharded) on a real GPU: two
processes (gloo for the boundary rows; both use cuda:0 here -- on a node each rank has its own GPU), each solving its frame
range chunk-parallel, stitched to the sequential chain."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, outdir, F):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from moshpp_amd import workload
    from moshpp_amd.parallel import solve_sequence_sharded, frame_ranges
    job = workload.make_job('smplh', F, 53, seed=1000)          # every rank builds the same seeded sequence
    solver = workload.make_solver(job)
    calls = []

    def solve_range(a, b, init):
        calls.append((a, b, init is not None))
        return solver.solve(job['obs'][a:b], job['vis'][a:b], chain_mode='chunked', verify_tol=1e-10, init=init)

    out, info = solve_sequence_sharded(solve_range, F, dist=dist, warmup=32, verify_tol=1e-9)
    a, b = frame_ranges(F, world)[rank]
    seq = solver.solve(job['obs'], job['vis'])                   # the literal sequential chain (each rank checks its part)
    dp = float(np.abs(out['fullpose'] - seq['fullpose'][a:b]).max())
    dt = float(np.abs(out['trans'] - seq['trans'][a:b]).max())
    res = [None] * world
    dist.all_gather_object(res, dict(rank=rank, calls=calls, info=info, dp=dp, dt=dt))
    if rank == 0:
        np.save(os.path.join(outdir, 'res.npy'), np.array(res, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def test_one_sequence_over_two_ranks_equals_sequential_chain(gpu_lib, tmp_path):
    F = 1200
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), F), nprocs=2, join=True)
    res = np.load(tmp_path / 'res.npy', allow_pickle=True)
    for r in res:
        print(r)
        assert r['dp'] < 1e-7 and r['dt'] < 1e-7                 # stitched == sequential chain on every owned frame
        assert r['info']['max_handoff_dev'] <= 1e-9
    assert res[0]['calls'] == [(0, 600, False)]
    assert res[1]['calls'][0] == (600 - 32, 1200, False)


# ---- Stage-I: the picked frames of one subject over ranks, normal equations all-reduced ---------------------------------------
def _stagei_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from moshpp_amd import capi
    from moshpp_amd.parallel import stagei_solve_sharded
    from tests import helpers
    c = helpers.stagei_case(seed=12)
    mdl = c['model']
    dev = capi.Model(mdl['v_template'], mdl['shapedirs'], mdl['posedirs'], mdl['weights'], mdl['J_regressor'], mdl['parents'],
                     mdl['body_dof'], mdl['hand_dof'], mdl['hands_mean'], mdl['selected_components'])
    pr = capi.Prior(c['prior']['means'], c['prior']['chols'], c['prior']['weights'])
    kw = helpers.stagei_kwargs(c)
    out = stagei_solve_sharded(lambda **sh: capi.stagei_solve_host(dev, pr, **kw, **sh), len(c['frames']), dist)
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), betas=out['betas'], markers_latent=out['markers_latent'], pose=out['pose'],
             trans=out['trans'], iters=out['iters'])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('solver', ['schur', 'dense'])
def test_stagei_frames_over_two_ranks_equals_single_process(gpu_lib, tmp_path, solver, monkeypatch):
    """Two processes on the GPU, the picked frames split between them.  'schur' (the default arrow-structured solver): every rank
    eliminates its own frames and only the Schur system of the shared block (+ a few n-vectors) is all-reduced; 'dense'
    (MOSHII_S1_SOLVER=dense): the whole normal equations are.  Both must reproduce the single-process solve."""
    monkeypatch.setenv('MOSHII_S1_SOLVER', solver)      # (inherited by the spawned ranks)
    from moshpp_amd import capi
    from tests import helpers
    mp.spawn(_stagei_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    c = helpers.stagei_case(seed=12)
    mdl = c['model']
    dev = capi.Model(mdl['v_template'], mdl['shapedirs'], mdl['posedirs'], mdl['weights'], mdl['J_regressor'], mdl['parents'],
                     mdl['body_dof'], mdl['hand_dof'], mdl['hands_mean'], mdl['selected_components'])
    pr = capi.Prior(c['prior']['means'], c['prior']['chols'], c['prior']['weights'])
    single = capi.stagei_solve_host(dev, pr, **helpers.stagei_kwargs(c))
    outs = [np.load(tmp_path / f'rank{r}.npz') for r in range(2)]
    for o in outs:
        assert int(o['iters']) == single['iters']
        assert np.abs(o['betas'] - single['betas']).max() < 1e-8 and np.abs(o['markers_latent'] - single['markers_latent']).max() < 1e-9
        assert np.abs(o['pose'] - single['pose']).max() < 1e-8 and np.abs(o['trans'] - single['trans']).max() < 1e-9
    assert np.array_equal(outs[0]['betas'], outs[1]['betas']) and np.array_equal(outs[0]['pose'], outs[1]['pose'])
