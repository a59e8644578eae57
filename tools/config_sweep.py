"""Sequential vs chunked Stage-II on the other BASELINE config shapes (GPU box): parity between the two modes + timing."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
import torch
from moshpp_amd import capi, workload
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
out = []
for name, kw in [('config0 SMPL 41 mk', dict(model_type='smpl', n_markers=41, optimize_fingers=False)),
                 ('config1 SMPL-H 53 mk body', dict(model_type='smplh', n_markers=53, optimize_fingers=False)),
                 ('SMPL-H 73 mk + fingers', dict(model_type='smplh', n_markers=73, optimize_fingers=True)),
                 ('config2 SMPL-X 89 mk + fingers (no face)', dict(model_type='smplx', n_markers=89, optimize_fingers=True)),
                 ('config3 MANO 33 mk', dict(model_type='mano', n_markers=33, optimize_fingers=True))]:
    job = workload.make_job(n_frames=F, seed=2000, **kw)
    solver = workload.make_solver(job)
    ds = workload.DeviceSequence(job, solver, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ds.solve_sequential(stream); torch.cuda.synchronize(); ts = time.perf_counter() - t0
    seq = ds.results()
    ds.solve_chunked(stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rep = ds.solve_chunked(stream); torch.cuda.synchronize(); tc = time.perf_counter() - t0
    ch = ds.results()
    ok = seq['status'] != 1
    dp = np.abs(seq['fullpose'] - ch['fullpose'])[ok].max(1)
    r = dict(config=name, frames=F, n_free=3 + len(solver.ids['step2']), kernel=capi.last_launch_info()[0], lds=capi.last_launch_info()[1],
             sequential_fps=round(F / ts, 1), sequential_us_per_frame=round(ts / F * 1e6, 1), chunked_fps=round(F / tc, 1),
             chunked_vs_sequential_max_rad=float(dp.max()), frames_over_1e_9=int((dp > 1e-9).sum()), iters_per_frame=round(float(seq['iters'][ok, 0].mean()), 2),
             failed_solves=int((seq['status'] < 0).sum()), **rep)
    print(json.dumps(r), flush=True)
    out.append(r)
json.dump(out, open('gpurun_out/config_sweep.json', 'w'), indent=1)
