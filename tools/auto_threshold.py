"""Where mosh_stageii's default chain mode ('auto', chmosh.StageIISolver.choose_chain_mode) switches from the sequential (cooperative)
chain to the chunked solve: both timed through the host-buffer path mosh_stageii uses, on the first F frames of the bench sequences.
    python tools/auto_threshold.py [--frames=64,128,256,512,1024,2048,4000] [--seeds=1000,123,71]  ->  profiles/r05_auto_threshold.txt"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import workload
opt = {a.split('=')[0]: a.split('=')[1] for a in sys.argv[1:] if a.startswith('--') and '=' in a}
Fs = [int(x) for x in opt.get('--frames', '64,128,256,512,1024,2048,4000').split(',')]
seeds = [int(x) for x in opt.get('--seeds', '1000,123,71').split(',')]
print('# ms per solve through host buffers (best of 3), first F frames of the SMPL-H / 53-marker bench sequences; sequential = the cooperative chain')
for sd in seeds:
    job = workload.make_job('smplh', n_frames=max(Fs), n_markers=53, seed=sd)
    solver = workload.make_solver(job)
    for F in Fs:
        obs, vis = job['obs'][:F], job['vis'][:F]
        t = {}
        for mode in ('sequential', 'chunked'):
            solver.solve(obs, vis, chain_mode=mode)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); out = solver.solve(obs, vis, chain_mode=mode); best = min(best, time.perf_counter() - t0)
            t[mode] = (best, out)
        d = np.abs(t['chunked'][1]['fullpose'] - t['sequential'][1]['fullpose']).max()
        print(f'seed {sd:5d} F {F:5d}: sequential {t["sequential"][0] * 1e3:8.2f} ms   chunked {t["chunked"][0] * 1e3:8.2f} ms   ratio {t["sequential"][0] / t["chunked"][0]:5.2f}'
              f'   auto picks {solver.choose_chain_mode(F):10s}  max|dpose| {d:.1e}', flush=True)
