"""Chunked solve with EXACT hand-offs (verify_tol -> 0: only bitwise-equal states are accepted) for several warm-up lengths.
python tools/exact_mode.py [seed ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from moshpp_amd import workload
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
seeds = [int(a) for a in sys.argv[1:]] or [1000, 123]
for seed in seeds:
    job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=seed)
    solver = workload.make_solver(job)
    ds = workload.DeviceSequence(job, solver, dev)
    ds.solve_sequential(stream); torch.cuda.synchronize()
    seq = ds.results()
    for W, tol in ((32, 1e-9), (32, 1e-300), (48, 1e-300), (64, 1e-300), (80, 1e-300)):
        ds.solve_chunked(stream, warmup=W, verify_tol=tol)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            rep = ds.solve_chunked(stream, warmup=W, verify_tol=tol)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 2 * 1e3
        chk = ds.results()
        dp = np.abs(chk['fullpose'] - seq['fullpose']).max(1)
        print(f'seed {seed} W={W} tol={tol:g}: {ms:.1f} ms/step, repaired {rep["n_repaired"]}/{rep["n_chunks"]} in {rep["repair_rounds"]} rounds, '
              f'max dev {dp.max():.2e} rad, frames > 1e-4: {(dp > 1e-4).sum()}, > 1e-9: {(dp > 1e-9).sum()}', flush=True)
