#!/bin/bash
export PYTHONPATH=. TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
cat > /tmp/clk.py <<'P'
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from moshpp_amd import capi, workload
lib = capi.load()
buf = (C.c_longlong * 64)()
lib.moshii_prof_read.argtypes = [C.POINTER(C.c_longlong), C.c_int]
dev = torch.device('cuda', 0); stream = torch.cuda.current_stream().cuda_stream
job = workload.make_job('smplh', 4000, 53, seed=1000); solver = workload.make_solver(job)
ds = workload.DeviceSequence(job, solver, dev)
for name, fn in (('one sequential cooperative chain (6 CUs busy)', lambda: ds.solve_sequential(stream)),
                 ('one sequential one-workgroup chain (1 CU busy)', lambda: ds.solve_sequential(stream, coop=1)),
                 ('chunked solve: pass 1 = 250 chains, one per CU (block 0 = the first chunk chain)', lambda: ds.solve_chunked(stream, verify_tol=1e-9, coop=1))):
    fn(); torch.cuda.synchronize(); lib.moshii_prof_read(buf, 1)
    fn(); torch.cuda.synchronize(); lib.moshii_prof_read(buf, 1)
    p = np.array(list(buf), dtype=np.float64)
    print(f'{name}: shader clock of block 0 during its kernel(s): {p[12] / (p[30] / 100e6) / 1e6:.0f} MHz', flush=True)
P
MOSHII_LIB=$PWD/moshpp_amd/libmoshii_prof.so python /tmp/clk.py 2>&1 | grep -v amdgpu.ids > $O/chain_clock.txt
MOSHII_LIB=$PWD/moshpp_amd/libmoshii_prof.so python tools/prof_chain.py 400 smplh 2>&1 | grep -v amdgpu.ids > $O/prof_coop.txt
cat $O/chain_clock.txt; head -30 $O/prof_coop.txt
