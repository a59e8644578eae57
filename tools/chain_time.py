"""Wall time of one sequential chain (host buffers): python tools/chain_time.py [F=400] [model=smplh] [fingers]"""
import sys
import time
sys.path.insert(0, '.')
from moshpp_amd import capi, workload
F = int(sys.argv[1]) if len(sys.argv) > 1 else 400
mt = sys.argv[2] if len(sys.argv) > 2 else 'smplh'
fingers = len(sys.argv) > 3 and sys.argv[3] == 'fingers'
M = {'smplh': 53, 'smpl': 41, 'smplx': 89, 'mano': 33}[mt]
job = workload.make_job(mt, F, M, seed=1000, optimize_fingers=fingers)
solver = workload.make_solver(job)
solver.solve(job['obs'][:8], job['vis'][:8])
best = 1e9
for _ in range(3):
    t = time.perf_counter(); out = solver.solve(job['obs'], job['vis']); best = min(best, time.perf_counter() - t)
print(f'{mt} F={F} fingers={fingers}: {best / F * 1e6:.1f} us/frame, kernel {capi.last_launch_info()}, iters/frame {out["iters"][:, 0].mean():.2f}')
