"""Chunked-mode sweep #2 (GPU box): throughput / repairs / deviation from the sequential chain vs (C, W, tol, LDS budget)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from moshpp_amd import capi, workload

F = int(os.environ.get('EXPLORE_F', 4000))
dev = torch.device('cuda', 0)
job = workload.make_job('smplh', n_frames=F, n_markers=53, seed=1000)
solver = workload.make_solver(job)
ds = workload.DeviceSequence(job, solver, dev)
stream = torch.cuda.current_stream().cuda_stream
runs = []


def timed(fn, reps=2, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def setenv(**kw):
    for k in ('MOSHII_LDS_BUDGET', 'MOSHII_TM', 'MOSHII_TWO_PER_CU'):
        os.environ.pop(k, None)
    for k, v in kw.items():
        if v is not None:
            os.environ[k] = str(v)


setenv()
t = timed(lambda: ds.solve_sequential(stream), reps=1, warm=0)
ref = ds.results()
r = dict(kind='sequential', ms=t * 1e3, fps=F / t, lds=capi.last_launch_info()[1])
print(r, flush=True); runs.append(r)
solved = ref['status'] == 0
sweep = json.loads(os.environ.get('EXPLORE_SWEEP', 'null')) or [
    (two, C, W, tol) for two in (0, 1) for C in (256, 512) for (W, tol) in ((32, 1e-9), (24, 1e-8))]
for budget, C, W, tol in sweep:
    setenv(MOSHII_TWO_PER_CU=budget)
    try:
        t = timed(lambda: ds.solve_chunked(stream, num_chunks=C, warmup=W, verify_tol=tol), reps=2, warm=1)
        out = ds.results()
        dp = np.abs(out['fullpose'] - ref['fullpose'])[solved].max(1)
        dm = float(np.abs(out['markers_sim'][solved] - ref['markers_sim'][solved]).max())
        r = dict(kind='chunked', budget=budget, C=C, W=W, tol=tol, lds=capi.last_launch_info()[1], ms=t * 1e3, fps=F / t,
                 max_dpose=float(dp.max()), frames_over_1e4=int((dp > 1e-4).sum()), frames_over_1e6=int((dp > 1e-6).sum()),
                 max_dmarker=dm, status_equal=bool((out['status'] == ref['status']).all()), **ds.report)
    except Exception as e:
        r = dict(kind='chunked', budget=budget, C=C, W=W, error=repr(e))
    print(r, flush=True); runs.append(r)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(runs, open('gpurun_out/explore2.json', 'w'), indent=1)
