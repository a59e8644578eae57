"""Writes tests/golden/stagei_golden.npz: the Stage-I oracle's solution of two seeded problems (helpers.stagei_case defaults, and the
finger variant), so that the oracle itself is regression-pinned and the GPU tests can compare against committed numbers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import stagei_oracle as s1          # noqa: E402
from tests import helpers                       # noqa: E402

CASES = {'body': (dict(), False), 'fingers': (dict(finger_markers=True, M=36, seed=2), True)}

if __name__ == '__main__':
    out = {}
    for name, (kw, fingers) in CASES.items():
        c = helpers.stagei_case(**kw)
        ref = s1.stagei_solve(c['m'], c['faces'], c['prior'], 'smplh', c['frames'], c['vids'], c['mask'], c['m2b'], c['nb'],
                              optimize_fingers=fingers)
        for k in ('betas', 'markers_latent', 'pose', 'trans', 'markers_latent_vids'):
            out[f'{name}_{k}'] = ref[k]
        out[f'{name}_errs'] = np.array([ref['errs'][k] for k in sorted(ref['errs'])])
        out[f'{name}_err_names'] = np.array(sorted(ref['errs']))
    fn = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'stagei_golden.npz')
    np.savez_compressed(fn, **out)
    print(fn, os.path.getsize(fn))
