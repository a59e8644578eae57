"""One full 4000-frame capture of the BASELINE config-3 subject (capture 7000 of tests/golden/make_config3_golden.py's generator) through
the NumPy oracle and through the GPU chain, plain and cooperative.
    python tools/config3_full_parity.py oracle      (CPU, ~6 min: writes tools/_config3_oracle_7000.npz)
    python tools/config3_full_parity.py gpu         (GPU box, with that file present: prints the comparison)"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import workload
from tests.helpers import face_capture_host, face_job_oracle
F, MS = 4000, 7000
job = workload.make_face_job()
m, pr, closest, coef = face_job_oracle(job)
cap = face_capture_host(job, m, closest, coef, MS, F)
if sys.argv[1] == 'oracle':
    from oracle import stageii_oracle as so
    t0 = time.time()
    ref = so.stageii_chain(m, pr, closest, coef, cap['obs'], cap['vis'], 'smplx', optimize_fingers=True, optimize_face=True, free_shape='expr')
    print('oracle:', time.time() - t0, 's')
    np.savez_compressed('tools/_config3_oracle_7000.npz', fullpose=ref['fullpose'], shape=ref['shape'], trans=ref['trans'], iters=np.asarray(ref['iters']),
                        data_sse=np.asarray(ref['errs']['data']))
else:
    from moshpp_amd import capi
    ref = np.load('tools/_config3_oracle_7000.npz')
    solver = workload.make_solver(job)
    ch = [dict(attach=solver.attach, obs=cap['obs'], vis=cap['vis'], first=True)]
    print(f'BASELINE config-3 subject, capture {MS}, {F} frames, 89 markers, 194 unknowns in Step 2; oracle data SSE: median {np.median(ref["data_sse"]):.2f}, '
          f'frames > 100: {np.flatnonzero(ref["data_sse"] > 100).tolist()}')
    for coop, name in ((1, 'one workgroup'), (0, 'cooperative (library default)')):
        t0 = time.perf_counter()
        o = capi.chain_solve_host(solver.dev, solver.prior, solver.opts, ch, coop=coop)[0]
        dt = time.perf_counter() - t0
        dp = np.abs(o['fullpose'] - ref['fullpose']).max(1)
        ds = np.abs(o['shape'] - ref['shape']).max(1)
        same = o['iters'][:, 0] == ref['iters']
        first = int(np.argmax(dp > 1e-6)) if (dp > 1e-6).any() else -1
        d = (o['markers_sim'] - cap['obs'])[cap['vis']]
        print(f'{name} ({capi.last_launch_info()[0]}, {dt / F * 1e3:.2f} ms/frame): vs oracle max|dpose| {dp.max():.2e} rad, max|dexpr| {ds.max():.2e}; frames > 1e-6 rad: {(dp > 1e-6).sum()} '
              f'(first: {first}), > 1e-4 rad: {(dp > 1e-4).sum()}; before the first: max {dp[:first if first > 0 else F].max():.2e}; iteration counts equal on {same.mean() * 100:.2f} % of frames; '
              f'status != 0 on {(o["status"] != 0).sum()} frames; marker rmse {np.sqrt((d ** 2).sum(1).mean()) * 1e3:.2f} mm')
