"""The chunked solve of the bench sequences with its sweeps as plain chains carried on inside the first launch (the default) and as
cooperative chains launched by the host's rounds (MOSHII_COOP_GROUP(g)); and the sequential chain plain / cooperative.
    python tools/chunk_modes.py [--groups=0,4,6,8] [--seeds=1000,123,71,5,2024,7] [--frames=4000] [--reps=3]"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import torch
from moshpp_amd import capi, workload
opt = {a.split('=')[0]: a.split('=')[1] for a in sys.argv[1:] if a.startswith('--') and '=' in a}
groups = [int(x) for x in opt.get('--groups', '0,4,6,8').split(',')]
seeds = [int(x) for x in opt.get('--seeds', '1000,123,71,5,2024,7').split(',')]
F = int(opt.get('--frames', 4000))
reps = int(opt.get('--reps', 3))
tol = float(opt.get('--verify-tol', 1e-9))
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
tot = {g: 0.0 for g in groups}
seqt = {0: 0.0, 6: 0.0}
for sd in seeds:
    job = workload.make_job('smplh', n_frames=F, n_markers=53, seed=sd)
    solver = workload.make_solver(job)
    ds = workload.DeviceSequence(job, solver, dev)
    ds.solve_sequential(stream); torch.cuda.synchronize()
    t0 = time.perf_counter(); ds.solve_sequential(stream); torch.cuda.synchronize(); ts = time.perf_counter() - t0
    ref = ds.results()
    t0 = time.perf_counter(); ds.solve_sequential(stream, coop=6); torch.cuda.synchronize(); tc = time.perf_counter() - t0
    rc = ds.results()
    ok = ref['status'] == 0
    dps = np.abs(rc['fullpose'] - ref['fullpose'])[ok].max(1)
    seqt[0] += ts; seqt[6] += tc
    line = (f'seed {sd}: sequential plain {ts / F * 1e6:6.1f} us/frame, cooperative(6) {tc / F * 1e6:6.1f} us/frame, max|dpose| {dps.max():.1e} '
            f'(frames > 1e-4: {(dps > 1e-4).sum()}, > 1e-9: {(dps > 1e-9).sum()}), iters equal on {(rc["iters"] == ref["iters"]).all(1).mean() * 100:.2f} % of frames')
    print(line, flush=True)
    for g in groups:
        ds.solve_chunked(stream, verify_tol=tol, coop=g); torch.cuda.synchronize()
        best, rp = 1e9, None
        for _ in range(reps):
            t0 = time.perf_counter(); rp = ds.solve_chunked(stream, verify_tol=tol, coop=g); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        r = ds.results()
        dp = np.abs(r['fullpose'] - ref['fullpose'])[ok].max(1)
        tot[g] += best
        print(f'    chunked, sweeps {"plain, carried on" if g == 0 else f"cooperative({g}), host rounds"}: {best * 1e3:7.2f} ms  ({F / best / 1e3:6.1f} k frames/s)  repaired {rp["n_repaired"]} in {rp["repair_rounds"]} rounds'
              f'  max|dpose| vs sequential {dp.max():.1e} (frames > 1e-4: {(dp > 1e-4).sum()}, > 1e-9: {(dp > 1e-9).sum()})', flush=True)
n = len(seeds) * F
print(f'sequential: plain {n / seqt[0]:.0f} frames/s, cooperative(6) {n / seqt[6]:.0f} frames/s')
for g in groups:
    print(f'chunked aggregate, sweeps {"plain" if g == 0 else f"cooperative({g})"}: {n / tot[g] / 1e3:.1f} k frames/s ({tot[g] * 1e3:.1f} ms for {len(seeds)} sequences)')
