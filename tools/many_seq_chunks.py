"""A rank's share of the fixed many-sequence job at N = 8 (32 distinct 4000-frame captures of one subject) on one GPU: time per pass and
repair statistics against the number of chunks per sequence, and with the repair rounds traced (MOSHII_TRACE_REPAIR=1 prints every round).
python tools/many_seq_chunks.py [n_sequences=32] [chunk counts ...]   (hand-offs verified to the product default, 1e-9)"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from moshpp_amd import workload
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 32
chunks = [int(a) for a in sys.argv[2:]] or [8, 12, 16]
job = workload.make_job('smplh', 4000, 53, seed=1000)
solver = workload.make_solver(job)
dev = torch.device('cuda:0')
stream = torch.cuda.current_stream().cuda_stream
copies = [workload.DeviceSequence(workload.make_capture(job, solver, 5000 + i), solver, dev) for i in range(NS)]
for nc in chunks:
    workload.solve_many_chunked(copies, stream, num_chunks=nc, verify_tol=1e-9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = workload.solve_many_chunked(copies, stream, num_chunks=nc, verify_tol=1e-9)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{NS} sequences x {nc} chunks: {dt * 1e3:.1f} ms = {NS * 4000 / dt / 1e3:.0f} k frames/s; {rep}')
