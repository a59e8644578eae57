import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from moshpp_amd import workload, capi
job = workload.make_job('smplh', 4000, 53, seed=1000)
solver = workload.make_solver(job)
def t(n, start=0, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        o = capi.chain_solve_host(solver.dev, solver.prior, solver.opts, [dict(attach=solver.attach, obs=job['obs'][start:start+n], vis=job['vis'][start:start+n], first=True)], coop=1)[0]
        best = min(best, time.perf_counter() - t0)
    return best, o['iters'][:min(n, 3)].tolist()
for start in (0, 1000, 2000, 3000):
    a, ia = t(1, start); b, ib = t(2, start); c, ic = t(33, start); d, _ = t(49, start)
    print(f'start {start}: 1 frame {a*1e3:.2f} ms (iters/evals {ia}), 2 frames {b*1e3:.2f} ms, 33 frames {c*1e3:.2f} ms, 49 frames {d*1e3:.2f} ms -> per later frame {(d-c)/16*1e6:.0f} us')
