"""Which hand-offs fail, and does marker visibility in the warm-up window explain it?  (GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from moshpp_amd import workload
dev = torch.device('cuda', 0)
job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=1000)
solver = workload.make_solver(job)
ds = workload.DeviceSequence(job, solver, dev)
stream = torch.cuda.current_stream().cuda_stream
vis = job['vis']
for W in (16, 32, 64):
    os.environ['MOSHII_DUMP_HANDOFF'] = f'gpurun_out/handoff_W{W}.txt'
    rep = ds.solve_chunked(stream, num_chunks=250, warmup=W, verify_tol=1e-9)
    d = np.loadtxt(f'gpurun_out/handoff_W{W}.txt')
    a, s, dev_ = d[:, 1].astype(int), d[:, 2].astype(int), d[:, 4]
    nv = np.array([vis[ai:si].sum(1).min() if si > ai else 53 for ai, si in zip(a, s)])
    mv = np.array([vis[ai:si].sum(1).mean() if si > ai else 53 for ai, si in zip(a, s)])
    ok = dev_ <= 1e-9
    print(f'W={W}: fail {int((~ok[1:]).sum())}/{len(ok)-1}; log10 dev percentiles 50/80/90/99: '
          + ' '.join(f'{np.log10(np.percentile(dev_[1:] + 1e-300, q)):.1f}' for q in (50, 80, 90, 99)))
    print('   min visible markers in warm-up window:  passing mean %.1f   failing mean %.1f' % (nv[1:][ok[1:]].mean(), nv[1:][~ok[1:]].mean()))
    print('   mean visible markers in warm-up window: passing mean %.1f   failing mean %.1f' % (mv[1:][ok[1:]].mean(), mv[1:][~ok[1:]].mean()))
    print('   corr(log dev, min vis) = %.2f' % np.corrcoef(np.log10(dev_[1:] + 1e-16), nv[1:])[0, 1], rep)
