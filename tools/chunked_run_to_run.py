"""Run-to-run variation of the chunked solve (mosh_stageii's default for long body-only captures): R repeated solves of every bench
sequence through the host-buffer path, against the first of them and against the sequential chain."""
import sys
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import workload
R = 12
print(f'# {R} repeated chunked solves (verify_tol 1e-9, the default) of each 4000-frame bench sequence; rad')
for seed in (1000, 123, 71, 5, 2024, 7):
    job = workload.make_job('smplh', 4000, 53, seed=seed)
    solver = workload.make_solver(job)
    seq = solver.solve(job['obs'], job['vis'], chain_mode='sequential')
    runs = [solver.solve(job['obs'], job['vis'], chain_mode='auto') for _ in range(R)]
    assert all(r['chain_mode'] == 'chunked' for r in runs)
    d_run = max(float(np.abs(r['fullpose'] - runs[0]['fullpose']).max()) for r in runs[1:])
    same = sum(1 for r in runs[1:] if np.array_equal(r['fullpose'], runs[0]['fullpose']))
    dv = np.stack([np.abs(r['fullpose'] - seq['fullpose']).max(1) for r in runs])
    print(f'seed {seed:5d}: max |run - first run| {d_run:.2e} ({same} of {R - 1} runs bit-identical to the first) | vs the sequential chain: max {dv.max():.2e}, '
          f'frames over 1e-7 in any run {int((dv.max(0) > 1e-7).sum())}, over 1e-4 {int((dv.max(0) > 1e-4).sum())} | repaired chunks per run {[int(r["chunk_report"]["n_repaired"]) for r in runs][:6]}', flush=True)
