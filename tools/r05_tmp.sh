#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
cat > /tmp/seqt.py <<'P'
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from moshpp_amd import capi, workload
F = 1500
dev = torch.device('cuda', 0); stream = torch.cuda.current_stream().cuda_stream
for sd in (1000,):
    job = workload.make_job('smplh', n_frames=F, n_markers=53, seed=sd)
    solver = workload.make_solver(job)
    ds = workload.DeviceSequence(job, solver, dev)
    for coop in (0, 1):
        ds.solve_sequential(stream, coop=coop); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); ds.solve_sequential(stream, coop=coop); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        r = ds.results(); print(f'{capi.last_launch_info()[0]:28s} checksum {float(np.abs(r["fullpose"]).sum()):.12f} {min(ts) / F * 1e6:7.2f} us/frame (median {sorted(ts)[2] / F * 1e6:7.2f})', flush=True)
    ts = []
    ds.solve_chunked(stream, verify_tol=1e-9, coop=1); torch.cuda.synchronize()
    for _ in range(5):
        t0 = time.perf_counter(); ds.solve_chunked(stream, verify_tol=1e-9, coop=1); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f'chunked, plain sweeps carried on: {min(ts) * 1e3:.2f} ms for {F} frames (median {sorted(ts)[2] * 1e3:.2f})', flush=True)
P
for rep in 1 2; do
for v in head new; do
  echo "== $v"; if [ $v = new ]; then python /tmp/seqt.py 2>&1 | grep -v amdgpu.ids; else MOSHII_LIB=$PWD/moshpp_amd/ab/libmoshii_$v.so python /tmp/seqt.py 2>&1 | grep -v amdgpu.ids; fi
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cooperative or free_shape or chain_kernel or sequence_solve_matches" 2>&1 | tail -3
