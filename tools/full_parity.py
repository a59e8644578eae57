"""GPU sequential chain and chunked solve over the full 4000-frame bench sequence vs the oracle's chain (tools/_oracle_full.npz,
made by tools/oracle_full.py on the CPU)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from moshpp_amd import workload
ref = np.load('tools/_oracle_full.npz')
dev = torch.device('cuda', 0)
job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=1000)
solver = workload.make_solver(job)
ds = workload.DeviceSequence(job, solver, dev)
stream = torch.cuda.current_stream().cuda_stream
ds.solve_sequential(stream)
seq = ds.results()
def cmp(name, a, b):
    d = np.abs(a - b).max(1)
    bad = np.flatnonzero(d > 1e-4)
    print(f'{name}: max {d.max():.3e} rad; frames > 1e-4: {len(bad)}; > 1e-6: {int((d > 1e-6).sum())}; > 1e-9: {int((d > 1e-9).sum())}; '
          f'first bad frames {bad[:8].tolist()}; median {np.median(d):.2e}')
    return d
assert (seq['status'] == 0).all() and len(ref['frame_ids']) == 4000
d1 = cmp('GPU sequential vs oracle', seq['fullpose'], ref['fullpose'])
print('   iteration counts equal on', int((seq['iters'][:, 0] == ref['iters']).sum()), 'of 4000 frames')
rep = ds.solve_chunked(stream)
ch = ds.results()
cmp('GPU chunked vs oracle   ', ch['fullpose'], ref['fullpose'])
cmp('GPU chunked vs GPU seq  ', ch['fullpose'], seq['fullpose'])
print(rep)
np.savez('gpurun_out/gpu_full.npz', seq=seq['fullpose'], chunked=ch['fullpose'], dev_seq_oracle=d1)
import os
os.environ['MOSHII_DUMP_HANDOFF'] = 'gpurun_out/handoff_full.txt'
rep = ds.solve_chunked(stream)
d = np.loadtxt('gpurun_out/handoff_full.txt')
for c in (114, 115, 116):
    print('chunk', c, 'launch_start/start/end', d[c, 1:4].astype(int).tolist(), 'first-pass hand-off dev %.3e' % d[c, 4])
for t in range(1838, 1846):
    print(t, 'iters seq', seq['iters'][t].tolist(), 'chunked', ch['iters'][t].tolist(), 'errs seq', seq['errs'][t].round(6).tolist(), 'chunked', ch['errs'][t].round(6).tolist(),
          'nvis', int(job['vis'][t].sum()))
# direct experiment: continue the sequential chain from its own state at 1839 as a separate chain
from moshpp_amd import capi
o = capi.chain_solve_host(solver.dev, solver.prior, solver.opts, [dict(attach=solver.attach, obs=job['obs'][1840:1850], vis=job['vis'][1840:1850], first=False,
                          init_pose=seq['pose'][1839], init_trans=seq['trans'][1839], init_pose_prev=seq['pose'][1838])])[0]
print('continuation from the sequential state: max dev vs seq', np.abs(o['fullpose'] - seq['fullpose'][1840:1850]).max(1))
