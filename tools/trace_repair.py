import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
from moshpp_amd import workload, capi
capi.load()
dev = torch.device('cuda', 0)
for seed in (123, 1000):
    job = workload.make_job('smplh', 4000, 53, seed=seed)
    solver = workload.make_solver(job)
    ds = workload.DeviceSequence(job, solver, dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): ds.solve_chunked(st, warmup=32, verify_tol=1e-9)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); rep = ds.solve_chunked(st, warmup=32, verify_tol=1e-9); torch.cuda.synchronize(); print(seed, 'ms', (time.perf_counter()-t0)*1e3, rep, flush=True)
