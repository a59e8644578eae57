#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
cat > /tmp/seqt.py <<'P'
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from moshpp_amd import capi, workload
F = 1500
dev = torch.device('cuda', 0); stream = torch.cuda.current_stream().cuda_stream
for sd in (1000, 123):
    job = workload.make_job('smplh', n_frames=F, n_markers=53, seed=sd)
    solver = workload.make_solver(job)
    ds = workload.DeviceSequence(job, solver, dev)
    for coop in (0, 1):
        ds.solve_sequential(stream, coop=coop); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); ds.solve_sequential(stream, coop=coop); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        r = ds.results()
        print(f'seed {sd}: {capi.last_launch_info()[0]} {best / F * 1e6:7.2f} us/frame  checksum {float(np.abs(r["fullpose"]).sum()):.12f}', flush=True)
    best = 1e9
    ds.solve_chunked(stream, verify_tol=1e-9); torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter(); ds.solve_chunked(stream, verify_tol=1e-9); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f'seed {sd}: chunked {best * 1e3:.2f} ms for {F} frames', flush=True)
P
python /tmp/seqt.py 2>&1 | grep -v amdgpu.ids
python bench.py --no-cpu --no-stagei --no-config3 --no-strong --no-sequential --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('bench value', d['value'], 'seeds', {k:v for k,v in d.get('plain_sweeps',{}).items() if k!='note'}, [ (s) for s in d['seeds']][:0])"
