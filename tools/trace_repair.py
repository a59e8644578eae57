"""Diagnostics: the repair rounds of one chunked solve (MOSHII_TRACE_REPAIR=1: chunk : hand-off deviation : frames the sweep ran / its limit;
a negative count: taken over by an upstream sweep at that frame), its time untraced, and its deviation from the sequential chain.
usage: python tools/trace_repair.py [seed ...]   (default 123 1000)"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from moshpp_amd import workload, capi
capi.load()
dev = torch.device('cuda', 0)
quiet = os.environ.get('TRACE_QUIET') is not None
for seed in ([int(a) for a in sys.argv[1:]] or [123, 1000]):
    job = workload.make_job('smplh', 4000, 53, seed=seed)
    solver = workload.make_solver(job)
    ds = workload.DeviceSequence(job, solver, dev)
    st = torch.cuda.current_stream().cuda_stream
    os.environ.pop('MOSHII_TRACE_REPAIR', None)
    ds.solve_sequential(st); torch.cuda.synchronize()
    ref = ds.results()
    for _ in range(2): ds.solve_chunked(st, warmup=32, verify_tol=1e-9)
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        t0 = time.perf_counter(); rep = ds.solve_chunked(st, warmup=32, verify_tol=1e-9); torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    got = ds.results()
    d = np.maximum(np.abs(got['pose'] - ref['pose']).max(1), np.abs(got['trans'] - ref['trans']).max(1))
    print(seed, 'ms', ' '.join(f'{m:.2f}' for m in ms), rep, '| vs sequential: max', f'{d.max():.2e}', 'frames over 1e-6:', int((d > 1e-6).sum()),
          'iters equal:', bool((got['iters'] == ref['iters']).all()), flush=True)
    if not quiet:
        os.environ['MOSHII_TRACE_REPAIR'] = '1'
        ds.solve_chunked(st, warmup=32, verify_tol=1e-9); torch.cuda.synchronize()
