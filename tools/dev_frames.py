"""Which frames of a chunk-parallel solve differ from the sequential chain?  python tools/dev_frames.py [seed] [tol]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from moshpp_amd import workload, capi
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 123
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-11
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=seed)
solver = workload.make_solver(job)
ds = workload.DeviceSequence(job, solver, dev)
ds.solve_sequential(stream); torch.cuda.synchronize()
seq = ds.results()
rep = ds.solve_chunked(stream, verify_tol=tol); torch.cuda.synchronize()
chk = ds.results()
dp = np.abs(chk['fullpose'] - seq['fullpose']).max(1)
bad = np.flatnonzero(dp > 1e-9)
starts, launch = capi.plan_chunks(4000, 250, 32)
print(rep)
print('bad frames', bad.tolist())
print('devs', [f'{d:.1e}' for d in dp[bad]])
print('chunk starts near', [int(s) for s in starts if bad.min() - 40 <= s <= bad.max() + 20])
print('nvis around', job['vis'][bad.min() - 3: bad.min() + 6].sum(1).tolist())
print('iters seq', seq['iters'][bad.min() - 2: bad.min() + 8].tolist())
print('iters chk', chk['iters'][bad.min() - 2: bad.min() + 8].tolist())
print('status seq', seq['status'][bad.min() - 2: bad.min() + 8].tolist(), 'chk', chk['status'][bad.min() - 2: bad.min() + 8].tolist())
print('errs seq', np.round(seq['errs'][bad.min() - 1: bad.min() + 3, :3], 6).tolist())
print('errs chk', np.round(chk['errs'][bad.min() - 1: bad.min() + 3, :3], 6).tolist())
