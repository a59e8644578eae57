"""How the config-3 workload (SMPL-X, 89 markers, fingers + jaw + 80 expression coefficients free) has to be generated so that the chain
TRACKS: for a few generator settings and motion seeds, 400 frames through the plain chain and through a cooperative chain of 8 --
two float64 executions of the same algorithm that differ in rounding only.  Where the problem is well conditioned they agree to
round-off over all frames (and so will the oracle); where the generator asks the solver for something the data cannot determine they
part ways.   python tools/config3_fixture.py [F=400]"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import capi, workload
F = int(sys.argv[1]) if len(sys.argv) > 1 else 400
settings = [dict(expr_boost=6.0, expr_amp=0.3, decay=1.0, vary=0.0),
            dict(expr_boost=6.0, expr_amp=0.6, decay=0.97, vary=0.0),
            dict(expr_boost=10.0, expr_amp=0.4, decay=1.0, vary=0.0),
            dict(expr_boost=10.0, expr_amp=0.4, decay=0.97, vary=0.5)]
if len(sys.argv) > 2:
    settings = [eval('dict(' + a + ')') for a in sys.argv[2:]]
for st in settings:
    job = workload.make_face_job(expr_boost=st['expr_boost'], expr_decay=st['decay'])
    solver = workload.make_solver(job)
    print('generator', st, flush=True)
    for ms in (7000, 7001, 7002, 7003, 7004):
        cap = workload.make_face_capture(job, solver, ms, n_frames=F, expr_amp=st['expr_amp'], expr_vary=st['vary'])
        ch = [dict(attach=solver.attach, obs=cap['obs'], vis=cap['vis'], first=True)]
        t0 = time.perf_counter(); a = capi.chain_solve_host(solver.dev, solver.prior, solver.opts, ch, coop=1)[0]; ta = time.perf_counter() - t0
        t0 = time.perf_counter(); b = capi.chain_solve_host(solver.dev, solver.prior, solver.opts, ch, coop=8)[0]; tb = time.perf_counter() - t0
        dp = np.abs(a['fullpose'] - b['fullpose']).max(1)
        ds = np.abs(a['shape'] - b['shape']).max(1)
        d = (a['markers_sim'] - cap['obs'])[cap['vis']]
        eg = cap['expr_gt'] if cap['expr_gt'].ndim == 2 else np.broadcast_to(cap['expr_gt'], a['shape'].shape)
        first_bad = int(np.argmax(dp > 1e-7)) if (dp > 1e-7).any() else -1
        print(f'  capture {ms}: plain {ta / F * 1e3:.2f} ms/frame, coop8 {tb / F * 1e3:.2f}; plain vs coop max|dpose| {dp.max():.1e} max|dexpr| {ds.max():.1e} (first frame > 1e-7: {first_bad}); '
              f'data SSE max {a["errs"][:, 0].max():.1f} median {np.median(a["errs"][:, 0]):.2f}; marker rmse {np.sqrt((d ** 2).sum(1).mean()) * 1e3:.2f} mm; '
              f'frames with data SSE > 100: {np.flatnonzero(a["errs"][:, 0] > 100)[:8]}; expr err max {np.abs(a["shape"] - eg).max():.2f} (first 10 coefficients {np.abs(a["shape"] - eg)[:, :10].max():.2f}); iters/frame {a["iters"][:, 0].mean():.2f}', flush=True)
