"""Chunk-repair trace of moshii_sequence_solve on the bench workload for a few seeds (MOSHII_TRACE_REPAIR=1 prints every round)."""
import os, sys, time
os.environ['MOSHII_TRACE_REPAIR'] = '1'
sys.path.insert(0, '.')
import torch
from moshpp_amd import capi, workload
capi.load()
dev = torch.device('cuda', 0)
for sd in [int(x) for x in sys.argv[1:]] or [7, 71]:
    job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=sd)
    solver = workload.make_solver(job)
    ds = workload.DeviceSequence(job, solver, dev)
    stream = torch.cuda.current_stream().cuda_stream
    print(f'==== seed {sd}', flush=True)
    sys.stdout.flush()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rep = ds.solve_chunked(stream, warmup=32, verify_tol=1e-9)
    torch.cuda.synchronize()
    print(f'seed {sd}: {1e3 * (time.perf_counter() - t0):.1f} ms', rep, flush=True)
