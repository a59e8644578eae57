"""Per-frame solve time at full occupancy: C independent chunks (no warm-up, no verification) of one long sequence,
vs the single-chain latency.  Tells how much of the single-chain time is clock/occupancy rather than the algorithm."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from moshpp_amd import capi, workload
L = int(os.environ.get('OCC_L', 120))
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
base = workload.make_job('smplh', n_frames=1024, n_markers=53, seed=1000)
solver = workload.make_solver(base)
for C in (1, 8, 32, 64, 128, 256, 512):
    F = C * L
    reps = (F + 1023) // 1024
    job = dict(base, obs=np.tile(base['obs'], (reps, 1, 1))[:F], vis=np.tile(base['vis'], (reps, 1))[:F])
    ds = workload.DeviceSequence(job, solver, dev)
    ds.solve_chunked(stream, num_chunks=C, warmup=0, verify_tol=1e300)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = ds.solve_chunked(stream, num_chunks=C, warmup=0, verify_tol=1e300)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    it = ds.iters.cpu().numpy()
    print(dict(C=C, kernel=capi.last_launch_info(), frames=F, ms=round(t * 1e3, 2), us_per_frame_per_chain=round(t * 1e6 / L, 1),
               fps=round(F / t), iters_per_frame=round(float(it[:, 0].mean()), 2), repaired=rep['n_repaired']), flush=True)
