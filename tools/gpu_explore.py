"""GPU exploration sweep (run on the MI355X box): sequential-chain latency vs LDS tiling, and the chunked mode's
throughput / deviation / repair rate as a function of (num_chunks, warmup, LDS budget).  Writes gpurun_out/explore.json."""
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
res = {'F': F, 'runs': []}


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
    for k in ('MOSHII_LDS_BUDGET', 'MOSHII_TM'):
        os.environ.pop(k, None)
    for k, v in kw.items():
        if v is not None:
            os.environ[k] = str(v)


# 1. sequential chain on a 400-frame prefix: LDS tiling sweep
ds400 = workload.DeviceSequence(dict(job, obs=job['obs'][:400], vis=job['vis'][:400]), solver, dev)
for tm in (2, 4, 8, 12, 16):
    setenv(MOSHII_LDS_BUDGET=160 * 1024, MOSHII_TM=tm)
    try:
        t = timed(lambda: ds400.solve_sequential(stream), reps=1, warm=1)
        name, lds, thr = capi.last_launch_info()
        r = dict(kind='sequential400', tm=tm, lds=lds, ms=t * 1e3, us_per_frame=t * 1e6 / 400)
    except Exception as e:
        r = dict(kind='sequential400', tm=tm, error=repr(e))
    print(r, flush=True); res['runs'].append(r)

# 2. full sequential reference (default settings)
setenv()
t = timed(lambda: ds.solve_sequential(stream), reps=1, warm=0)
ref = ds.results()
name, lds, thr = capi.last_launch_info()
r = dict(kind='sequential', ms=t * 1e3, fps=F / t, lds=lds)
print(r, flush=True); res['runs'].append(r)
solved = ref['status'] == 0

# 3. chunked sweep
for budget, tm in ((None, None), (80 * 1024, None), (120 * 1024, None), (160 * 1024, None)):
    for C in (128, 256, 512, 1024):
        for W in (8, 16):
            setenv(MOSHII_LDS_BUDGET=budget, MOSHII_TM=tm)
            try:
                t = timed(lambda: ds.solve_chunked(stream, num_chunks=C, warmup=W, verify_tol=1e-6), reps=2, warm=1)
                out = ds.results()
                name, lds, thr = capi.last_launch_info()
                dp = float(np.abs(out['fullpose'][solved] - ref['fullpose'][solved]).max())
                dm = float(np.abs(out['markers_sim'][solved] - ref['markers_sim'][solved]).max())
                r = dict(kind='chunked', budget=budget, tm=tm, C=C, W=W, lds=lds, ms=t * 1e3, fps=F / t, max_dpose=dp, max_dmarker=dm,
                         status_equal=bool((out['status'] == ref['status']).all()), **ds.report)
            except Exception as e:
                r = dict(kind='chunked', budget=budget, C=C, W=W, error=repr(e))
            print(r, flush=True); res['runs'].append(r)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/explore.json', 'w'), indent=1)
