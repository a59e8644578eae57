"""Frames where the chunk-parallel result differs from the sequential chain (iteration-count flips of the reference's e3 stop rule),
per seed and hand-off tolerance.  python tools/flip_sweep.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from moshpp_amd import workload
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
for seed in (1000, 123, 71, 5, 2024, 7):
    job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=seed)
    solver = workload.make_solver(job)
    ds = workload.DeviceSequence(job, solver, dev)
    ds.solve_sequential(stream); torch.cuda.synchronize()
    seq = ds.results()
    for tol in (1e-9, 1e-11, 1e-13):
        ds.solve_chunked(stream, verify_tol=tol)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rep = ds.solve_chunked(stream, verify_tol=tol)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        chk = ds.results()
        dp = np.abs(chk['fullpose'] - seq['fullpose']).max(1)
        it = (chk['iters'][:, 0] != seq['iters'][:, 0]).sum()
        print(f'seed {seed} tol={tol:g}: {ms:.1f} ms, repaired {rep["n_repaired"]} in {rep["repair_rounds"]} rounds, max dev {dp.max():.2e} rad, '
              f'frames > 1e-4: {(dp > 1e-4).sum()}, > 1e-9: {(dp > 1e-9).sum()}, iteration counts differ on {it} frames', flush=True)
