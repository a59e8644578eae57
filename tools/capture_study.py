"""Does a chain started from a STALE state of the sequential chain fall into the basin the sequential chain tracks?

For the chunk scheduler: a fresh (first-frame-schedule) start lands in another local minimum than the tracked one for ~10 % of
the starts, on some sequences for stretches of hundreds of frames -- the sweep that repairs such a stretch is sequential time.
Here: helper chains start at frame s from the sequential chain's state at frame s - 1 - J (a jump of J frames: pose, pose_prev,
trans of that frame), run `run` frames, and we record after how many frames they are within 1e-9 rad of the sequential chain.
    python tools/capture_study.py [seeds=1000,123,71,5,2024,7] [jumps=0,16,32,64,128] [run=64]
"""
import sys
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import capi, workload

seeds = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else '1000,123,71,5,2024,7').split(',')]
jumps = [int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else '0,16,32,64,128').split(',')]
RUN = int(sys.argv[3]) if len(sys.argv) > 3 else 64
F, STRIDE = 4000, 16
for seed in seeds:
    job = workload.make_job('smplh', F, 53, seed=seed)
    solver = workload.make_solver(job)
    seq = solver.solve(job['obs'], job['vis'])
    ok = seq['status'] == 0
    starts = [s for s in range(max(jumps) + 2, F - RUN, STRIDE)]
    for J in ['fresh'] + jumps:
        chains = []
        for s in starts:
            ch = dict(attach=solver.attach, obs=job['obs'][s:s + RUN], vis=job['vis'][s:s + RUN])
            if J == 'fresh':
                ch['first'] = True
            else:
                q = s - 1 - J
                ch.update(first=False, init_pose=seq['pose'][q], init_trans=seq['trans'][q], init_pose_prev=seq['pose'][q - 1])
            chains.append(ch)
        outs = capi.chain_solve_host(solver.dev, solver.prior, solver.opts, chains)
        conv = []     # frames until within 1e-9 of the sequential chain (and staying there for the rest of the run); RUN = never
        for s, o in zip(starts, outs):
            d = np.abs(o['fullpose'] - seq['fullpose'][s:s + RUN]).max(1)
            d[~ok[s:s + RUN]] = 0.0
            bad = np.flatnonzero(d > 1e-9)
            conv.append(int(bad[-1]) + 1 if len(bad) else 0)
        conv = np.array(conv)
        missed = np.flatnonzero(conv >= RUN)
        runs = np.split(missed, np.flatnonzero(np.diff(missed) > 1) + 1) if len(missed) else []
        print(f'seed {seed} jump {J!s:>5}: converged within 32 frames {np.mean(conv <= 32) * 100:5.1f} %, within 48 {np.mean(conv <= 48) * 100:5.1f} %, '
              f'never (in {RUN}) {len(missed):3d} of {len(starts)}; median frames {int(np.median(conv))}; '
              f'missed starts (frame ranges): {[(starts[r[0]], starts[r[-1]]) for r in runs][:8]}', flush=True)
