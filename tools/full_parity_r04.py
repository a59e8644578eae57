"""The full 4000-frame bench sequence (seed 1000) against the oracle's chain (tools/_oracle_full.npz, made by tools/oracle_full.py on
the CPU): the sequential chain as ONE workgroup, as the library runs it by default (a cooperative chain), and the chunked solve with
plain sweeps carried on inside the first launch and with cooperative sweeps (the default)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from moshpp_amd import capi, workload
ref = np.load('tools/_oracle_full.npz')
dev = torch.device('cuda', 0)
job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=1000)
solver = workload.make_solver(job)
ds = workload.DeviceSequence(job, solver, dev)
stream = torch.cuda.current_stream().cuda_stream
assert len(ref['frame_ids']) == 4000


def cmp(name, got, base, iters=None):
    d = np.abs(got['fullpose'] - base).max(1)
    line = (f'{name:64s}: max {d.max():.3e} rad; frames > 1e-4: {int((d > 1e-4).sum())}; > 1e-6: {int((d > 1e-6).sum())}; > 1e-9: {int((d > 1e-9).sum())}; median {np.median(d):.2e}')
    if iters is not None:
        line += f'; dogleg iteration counts equal on {int((got["iters"][:, 0] == iters).sum())} of 4000 frames'
    print(line, flush=True)


res = {}
for coop, name in ((1, 'sequential, one workgroup'), (0, 'sequential, library default')):
    ds.solve_sequential(stream, coop=coop); torch.cuda.synchronize()
    res[coop] = ds.results()
    assert (res[coop]['status'] == 0).all()
    cmp(f'{name} ({capi.last_launch_info()[0]}) vs oracle', res[coop], ref['fullpose'], ref['iters'])
cmp('sequential default vs sequential one workgroup', res[0], res[1]['fullpose'], res[1]['iters'][:, 0])
for coop, name in ((1, 'chunked, plain sweeps carried on in the first launch'), (0, 'chunked, library default (cooperative sweeps)')):
    rep = ds.solve_chunked(stream, coop=coop); torch.cuda.synchronize()
    ch = ds.results()
    cmp(f'{name} vs oracle', ch, ref['fullpose'])
    cmp(f'{name} vs sequential one workgroup', ch, res[1]['fullpose'])
    print('   ', rep)
