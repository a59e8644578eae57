"""CPU, world_size 2, gloo: the N>1 path (unit partition + result gather) of moshpp_amd.parallel."""
import os
import socket
import sys

import numpy as np
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
