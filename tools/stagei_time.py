"""Time moshii_stagei_solve on the bench's Stage-I workload (12 frames, 53 markers, 10 betas, SMPL-H-sized triangulated body).
Under rocprofv3 --kernel-trace --stats this gives the per-kernel split of one solve."""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moshpp_amd import capi        # noqa: E402
from moshpp_amd import workload    # noqa: E402


def main(reps=3, fingers=False):
    pb, dev, pr, kw = workload.make_stagei_job(optimize_fingers=fingers)
    for i in range(reps):
        t = time.perf_counter()
        out = capi.stagei_solve_host(dev, pr, **kw)
        print(f'rep {i}: {time.perf_counter() - t:.4f} s, {out["iters"]} iterations, errs {out["errs"]}', flush=True)


if __name__ == '__main__':
    main(reps=int(sys.argv[1]) if len(sys.argv) > 1 else 3, fingers='fingers' in sys.argv)
