"""Round 6: (a) the cost of a chain's first frame (first-frame schedule) against its later frames; (b) N copies of one 49-frame chain in ONE
launch, N = 1 .. 250 (identical chains: no slowest-chunk effect -- what is left is contention), and 250 DIFFERENT 49-frame chunks of the
sequence (the pass-1 launch of the chunked solve without its bookkeeping).  python tools/first_frame.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import workload, capi
job = workload.make_job('smplh', 4000, 53, seed=1000)
solver = workload.make_solver(job)


def run(chains, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        o = capi.chain_solve_host(solver.dev, solver.prior, solver.opts, chains, coop=1)
        best = min(best, time.perf_counter() - t0)
    return best, o


def ch(start, n):
    return dict(attach=solver.attach, obs=job['obs'][start:start + n], vis=job['vis'][start:start + n], first=True)


for start in (0, 1000, 2000, 3000):
    a, oa = run([ch(start, 1)]); b, _ = run([ch(start, 2)]); c, _ = run([ch(start, 33)]); d, _ = run([ch(start, 49)])
    print(f'start {start}: 1 frame {a*1e3:.2f} ms (iters/evals {oa[0]["iters"][:1].tolist()}), 2 frames {b*1e3:.2f} ms, 33 frames {c*1e3:.2f} ms, 49 frames {d*1e3:.2f} ms -> per later frame {(d-c)/16*1e6:.0f} us')
for n in (1, 8, 32, 64, 128, 250):
    t, _ = run([ch(1000, 49)] * n)
    print(f'{n:4d} copies of the 49-frame chain from frame 1000 in one launch: {t*1e3:.2f} ms (host staging included)')
t, outs = run([ch(16 * c, 49) for c in range(247)])
its = np.array([o['iters'][:, 0].sum() for o in outs])
print(f'247 different 49-frame chunks in one launch: {t*1e3:.2f} ms; dogleg iterations per chunk: min {its.min()} median {int(np.median(its))} max {its.max()}')
