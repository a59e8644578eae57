"""Fixture: the NumPy oracle's chain on captures of the BASELINE config-3 subject (workload.make_face_job: SMPL-X, 89 markers incl.
face / hand vertices, fingers + jaw + 80 expression coefficients free: 194 unknowns), 400 frames each, captures 7000, 7001, 7002
(tests.helpers.face_capture_host; nobody picked them: the first three motion seeds of the bench's config-3 leg).  Stored: the dogleg
iteration counts of EVERY frame, fullpose / expression / trans of every 10th frame.  ~0.15 s of CPU per frame.

    python tests/golden/make_config3_golden.py   ->  tests/golden/config3_oracle.npz
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from moshpp_amd import workload                      # noqa: E402
from oracle import stageii_oracle as so              # noqa: E402
from tests.helpers import face_capture_host, face_job_oracle   # noqa: E402

F, CAPS, STRIDE = 400, (7000, 7001, 7002), 10
job = workload.make_face_job()
m, pr, closest, coef = face_job_oracle(job)
out = dict(frames=F, stride=STRIDE, captures=np.array(CAPS))
for ms in CAPS:
    cap = face_capture_host(job, m, closest, coef, ms, F)
    t0 = time.time()
    ref = so.stageii_chain(m, pr, closest, coef, cap['obs'], cap['vis'], 'smplx', optimize_fingers=True, optimize_face=True, free_shape='expr')
    print(f'capture {ms}: {time.time() - t0:.0f} s, iterations per frame {ref["iters"].mean():.2f}, data SSE max {np.max(ref["errs"]["data"]):.1f}', flush=True)
    assert len(ref['frame_ids']) == F
    sel = np.arange(0, F, STRIDE)
    out[f'iters_{ms}'] = np.asarray(ref['iters'], dtype=np.int32)
    out[f'fullpose_{ms}'] = ref['fullpose'][sel]
    out[f'shape_{ms}'] = ref['shape'][sel]
    out[f'trans_{ms}'] = ref['trans'][sel]
    out[f'data_sse_{ms}'] = np.asarray(ref['errs']['data'])
np.savez_compressed(os.path.join(HERE, 'config3_oracle.npz'), **out)
print('wrote config3_oracle.npz')
