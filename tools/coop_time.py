"""Cooperative chains (MOSHII_COOP=g: g workgroups per chain) against the plain chain on the bench sequence: wall time per frame,
deviation, iteration counts.   python tools/coop_time.py [F=400] [model=smplh] [fingers] [--groups 2,3,4,5,6,8] [--fracs 0.2,0.4,0.7]"""
import os
import sys
import time
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import capi, workload
args = [a for a in sys.argv[1:] if not a.startswith('--')]
opt = {a.split('=')[0]: a.split('=')[1] for a in sys.argv[1:] if a.startswith('--') and '=' in a}
F = int(args[0]) if len(args) > 0 else 400
mt = args[1] if len(args) > 1 else 'smplh'
fingers = len(args) > 2 and args[2] == 'fingers'
groups = [int(x) for x in opt.get('--groups', '2,3,4,5,6,8').split(',')]
fracs = [float(x) for x in opt.get('--fracs', '0.2,0.4,0.7').split(',')]
M = {'smplh': 53, 'smpl': 41, 'smplx': 89, 'mano': 33}[mt]
job = workload.make_job(mt, F, M, seed=int(opt.get('--seed', 1000)), optimize_fingers=fingers)
solver = workload.make_solver(job)


def run(reps=3):
    solver.solve(job['obs'][:8], job['vis'][:8])
    best, out = 1e9, None
    for _ in range(reps):
        t = time.perf_counter(); out = solver.solve(job['obs'], job['vis']); best = min(best, time.perf_counter() - t)
    return best, out


os.environ['MOSHII_COOP'] = '1'   # plain chains (unset = the library's own choice)
t0, ref = run()
print(f'{mt} F={F} fingers={fingers} plain: {t0 / F * 1e6:7.1f} us/frame  {capi.last_launch_info()}  iters/frame {ref["iters"][:, 0].mean():.2f}', flush=True)
for g in groups:
    for fr in fracs:
        os.environ['MOSHII_COOP'] = str(g)
        os.environ['MOSHII_COOP_PRIOR_FRAC'] = str(fr)
        try:
            t, out = run()
        except Exception as e:
            print(f'  coop g={g} frac={fr}: FAILED {e!r}', flush=True)
            continue
        ok = out['status'] == 0
        dp = np.abs(out['fullpose'] - ref['fullpose'])[ok].max() if ok.any() else float('nan')
        same_it = bool((out['iters'] == ref['iters']).all())
        print(f'  coop g={g} prior_frac={fr}: {t / F * 1e6:7.1f} us/frame ({t0 / t:4.2f}x)  {capi.last_launch_info()[0]} lds {capi.last_launch_info()[1]}  '
              f'max|dpose| vs plain {dp:.2e}  iteration counts identical: {same_it}  status identical: {bool((out["status"] == ref["status"]).all())}', flush=True)
os.environ.pop('MOSHII_COOP', None)
os.environ.pop('MOSHII_COOP_PRIOR_FRAC', None)
