"""Dense vs arrow-structured Gauss-Newton solver of Stage-I (MOSHII_S1_SOLVER=dense / schur) on seeded problems: times and
differences.  `python tools/stagei_schur_check.py` runs the bench problem; `... seeds` a table over seeds / model families."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moshpp_amd import capi, workload     # noqa: E402


def compare(tag, **job):
    pb, dev, pr, kw = workload.make_stagei_job(**job)
    res, tm = {}, {}
    for mode in ('dense', 'schur'):
        os.environ['MOSHII_S1_SOLVER'] = mode
        capi.stagei_solve_host(dev, pr, **kw)
        t = time.perf_counter(); res[mode] = capi.stagei_solve_host(dev, pr, **kw); tm[mode] = time.perf_counter() - t
    a, b = res['dense'], res['schur']
    nb = len(a['betas'])
    print(f'{tag:34s} dense {tm["dense"]:.4f} s / {a["iters"]:3d} it   schur {tm["schur"]:.4f} s / {b["iters"]:3d} it   '
          f'max|dbetas| {np.abs(a["betas"] - b["betas"]).max() if nb else 0.0:.1e}  max|dmarkers_latent| '
          f'{np.abs(a["markers_latent"] - b["markers_latent"]).max():.1e}  max|dpose| {np.abs(a["pose"] - b["pose"]).max():.1e}  '
          f'vids equal {bool((a["markers_latent_vids"] == b["markers_latent_vids"]).all())}', flush=True)


if len(sys.argv) > 1 and sys.argv[1] == 'seeds':
    for sd in (1, 2, 3, 4, 5, 6):
        compare(f'smplh 53 mk 12 fr 10 betas seed {sd}', seed=sd)
    compare('smplh + fingers (24 dof/hand) seed 7', seed=7, optimize_fingers=True, n_markers=73)
    compare('smplx 10475 verts seed 8', model_type='smplx', n_verts=10475, seed=8)
    compare('smpl seed 9', model_type='smpl', seed=9)
    compare('smplh fixed betas seed 10', seed=10, nb=0)
else:
    compare('bench problem (seed 1)')
