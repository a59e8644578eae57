"""Dense vs arrow-structured (MOSHII_S1_SOLVER=schur) Gauss-Newton solver of Stage-I on the bench problem: times and differences."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moshpp_amd import capi, workload     # noqa: E402

pb, dev, pr, kw = workload.make_stagei_job()
res = {}
for mode in ('dense', 'schur'):
    if mode == 'schur':
        os.environ['MOSHII_S1_SOLVER'] = 'schur'
    else:
        os.environ.pop('MOSHII_S1_SOLVER', None)
    capi.stagei_solve_host(dev, pr, **kw)
    t = time.perf_counter(); o = capi.stagei_solve_host(dev, pr, **kw); dt = time.perf_counter() - t
    res[mode] = o
    print(f'{mode}: {dt:.4f} s, {o["iters"]} iterations', flush=True)
print('max|dbetas|', np.abs(res['dense']['betas'] - res['schur']['betas']).max(),
      'max|dmarkers_latent|', np.abs(res['dense']['markers_latent'] - res['schur']['markers_latent']).max(),
      'max|dpose|', np.abs(res['dense']['pose'] - res['schur']['pose']).max())
