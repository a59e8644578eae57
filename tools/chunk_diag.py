"""Where does a chunked solve differ from the sequential chain?  (seed, verify_tol) -> bad frame ranges + repair trace."""
import os, sys
os.environ['MOSHII_TRACE_REPAIR'] = '1'
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import workload
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 71
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-11
job = workload.make_job('smplh', 4000, 53, seed=seed)
solver = workload.make_solver(job)
seq = solver.solve(job['obs'], job['vis'])
for rep_i in range(3):
    chk = solver.solve(job['obs'], job['vis'], chain_mode='chunked', verify_tol=tol)
    d = np.abs(chk['fullpose'] - seq['fullpose']).max(1)
    bad = np.flatnonzero(d > 1e-7)
    print('run', rep_i, chk['chunk_report'], 'max dev', d.max(), 'bad frames', len(bad))
    if len(bad):
        runs = np.split(bad, np.flatnonzero(np.diff(bad) > 1) + 1)
        print(' bad ranges (frame: chunk of 16):', [(int(r[0]), int(r[-1]), int(r[0]) // 16, float(d[r].max())) for r in runs])
