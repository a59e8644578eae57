"""Chunked vs sequential over several seeds and repeats (the chunk scheme has run-to-run timing-dependent paths)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import workload
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-11
for seed in [int(x) for x in sys.argv[2:]] or [71, 7, 123, 5]:
    job = workload.make_job('smplh', 4000, 53, seed=seed)
    solver = workload.make_solver(job)
    seq = solver.solve(job['obs'], job['vis'])
    worst = 0.0
    for rep_i in range(int(__import__("os").environ.get("REPS", "8"))):
        chk = solver.solve(job['obs'], job['vis'], chain_mode='chunked', verify_tol=tol)
        d = np.abs(chk['fullpose'] - seq['fullpose']).max(1)
        worst = max(worst, float(d.max()))
        if d.max() > 1e-7:
            bad = np.flatnonzero(d > 1e-7)
            print('  seed', seed, 'run', rep_i, 'BAD frames', bad[0], '..', bad[-1], 'max', d.max(), chk['chunk_report'])
    print('seed', seed, 'worst over 8 runs', worst, flush=True)
