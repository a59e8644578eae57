"""Chunked solve time with the full first-frame schedule vs the lite warm-up start, over several seeded sequences.
python tools/lite_warmup_sweep.py  (run once per MOSHII_LITE_WARMUP setting)"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from moshpp_amd import workload
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
for seed in (1000, 71, 5, 123, 2024):
    job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=seed)
    solver = workload.make_solver(job)
    ds = workload.DeviceSequence(job, solver, dev)
    ds.solve_chunked(stream, verify_tol=1e-9)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        rep = ds.solve_chunked(stream, verify_tol=1e-9)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    chk = ds.results()
    ds.solve_sequential(stream); torch.cuda.synchronize()
    seq = ds.results()
    print(f'seed {seed}: {ms:.1f} ms/step, repaired {rep["n_repaired"]} in {rep["repair_rounds"]} rounds, '
          f'max|chunked - sequential| {np.abs(chk["fullpose"] - seq["fullpose"]).max():.2e} rad', flush=True)
