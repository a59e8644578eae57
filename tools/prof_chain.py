"""In-kernel phase timing of k_chain_solve (needs the -DMOSHII_PROFILE build: MOSHII_LIB=moshpp_amd/libmoshii_prof.so)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, '.')
from moshpp_amd import capi, workload
F = int(sys.argv[1]) if len(sys.argv) > 1 else 400
mt = sys.argv[2] if len(sys.argv) > 2 else 'smplh'
fingers = len(sys.argv) > 3 and sys.argv[3] == 'fingers'
M = {'smplh': 53, 'smpl': 41, 'smplx': 89, 'mano': 33}[mt]
job = workload.make_job(mt, F, M, seed=1000, optimize_fingers=fingers)
solver = workload.make_solver(job)
lib = capi.load()
buf = (C.c_longlong * 64)()
lib.moshii_prof_read.argtypes = [C.POINTER(C.c_longlong), C.c_int]
solver.solve(job['obs'][:8], job['vis'][:8])
lib.moshii_prof_read(buf, 1)
t = time.perf_counter(); out = solver.solve(job['obs'], job['vis']); dt = time.perf_counter() - t
lib.moshii_prof_read(buf, 1)
p = np.array(list(buf), dtype=np.float64)
names = {0: 'eval: fullpose/rodrigues/chain', 1: 'eval: posedirs', 2: 'eval: skin+markers', 3: 'eval: prior+reduce',
         4: 'asm: T0', 5: 'asm: T1 vertex jac', 6: 'asm: T2 marker rows', 7: 'asm: T3 JtJ', 8: 'asm: structured',
         9: 'chol: set-up + verdicts', 10: 'back-subst', 12: 'kernel total',
         46: 'chol: publish', 44: 'chol: panel (wave 0)', 45: 'chol: trailing update',
         40: 'eval: velocity / finger sums', 41: 'eval: prior setup (xb)', 42: 'eval: prior shortcut', 43: 'eval: prior full + argmin',
         31: 'coop: exchange after an evaluation (incl. waiting)', 32: 'coop: exchange of the normal equations (incl. waiting)'}
p[9] -= p[44] + p[45] + p[46]   # (slot 9 laps the whole factorisation; 44-46 are laps inside it)
tot = p[12]
print(f'{mt} F={F} fingers={fingers} wall {dt*1e3:.1f} ms  ({dt/F*1e6:.1f} us/frame)  launch {capi.last_launch_info()}')
print(f'evals {p[20]:.0f} ({p[20]/F:.2f}/frame) assembles {p[21]:.0f} ({p[21]/F:.2f}/frame) chol {p[22]:.0f} ({p[22]/F:.2f}/frame)')
us_per_tick = dt * 1e6 / tot
acc = 0
for k, nm in names.items():
    if k == 12: continue
    acc += p[k]
    print(f'  {nm:34s} {p[k]/tot*100:6.2f}%  {p[k]*us_per_tick/F:8.1f} us/frame')
print(f'  {"other (dogleg control, copies)":34s} {(tot-acc)/tot*100:6.2f}%  {(tot-acc)*us_per_tick/F:8.1f} us/frame')
ph = lambda ks: sum(p[k] for k in ks) * us_per_tick / F
print(f'  calls (thread 0, wall): eval_forward_fn {p[15]*us_per_tick/F:.1f} (phases inside {ph([0,1,2,3]):.1f}) | assemble_fn {p[16]*us_per_tick/F:.1f} (inside {ph([4,5,6,7,8]):.1f})'
      f' | ldl_solve {p[17]*us_per_tick/F:.1f} (inside {ph([9,10]):.1f}) | frame setup {p[18]*us_per_tick/F:.1f} | frame end {p[19]*us_per_tick/F:.1f} us/frame')
print(f'  control (thread 0, wall): post-eval {p[23]*us_per_tick/F:.1f} | gradient max {p[24]*us_per_tick/F:.1f} | radius + start_iteration {p[25]*us_per_tick/F:.1f}'
      f' | trial point {p[27]*us_per_tick/F:.1f} us/frame')
print(f'  back-substitution, thread 0 from entry to its last store: {p[35]*us_per_tick/F:.1f} us/frame')
if p[33] or p[34]:
    print(f'  coop (thread 0, wall): drain + barrier before the flag {p[33]*us_per_tick/F:.1f} | flag store .. every rank seen {p[34]*us_per_tick/F:.1f} us/frame')
print(f'  prior: shortcut taken {p[28]:.0f} times, full evaluation {p[29]:.0f} times')
print(f'shader clock during the kernel: {p[12] / (p[30] / 100e6) / 1e6:.0f} MHz (s_memtime ticks / s_memrealtime @100 MHz)')
print('iters/frame', out['iters'][:,0].mean(), 'status', np.unique(out['status'], return_counts=True))
