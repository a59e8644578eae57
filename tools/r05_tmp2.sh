#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "skew or cooperative_chain" 2>&1 | tail -5
cat > /tmp/rep.py <<'P'
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from moshpp_amd import capi, workload
F = 600
dev = torch.device('cuda', 0); stream = torch.cuda.current_stream().cuda_stream
job = workload.make_job('smplh', n_frames=F, n_markers=53, seed=1000)
solver = workload.make_solver(job)
ds = workload.DeviceSequence(job, solver, dev)
ds.solve_sequential(stream, coop=1); torch.cuda.synchronize(); p = ds.results()
outs = []
for i in range(5):
    ds.solve_sequential(stream, coop=0); torch.cuda.synchronize(); r = ds.results()
    outs.append(r['fullpose'].copy())
    print(i, capi.last_launch_info()[0], 'checksum %.15f' % float(np.abs(r['fullpose']).sum()), 'max|d| vs run 0 %.3e' % np.abs(outs[-1] - outs[0]).max(), 'vs plain %.3e' % np.abs(outs[-1] - p['fullpose']).max(), 'iters equal', bool((r['iters'] == p['iters']).all()))
P
python /tmp/rep.py 2>&1 | grep -v amdgpu.ids
