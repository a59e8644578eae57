"""Stage-I on the GPU (moshii_stagei_solve) against the f64 oracle (oracle/stagei_oracle.py) on the same seeded problems."""
import time

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def _device(case):
    from moshpp_amd import capi
    mdl = case['model']
    dev = capi.Model(mdl['v_template'], mdl['shapedirs'], mdl['posedirs'], mdl['weights'], mdl['J_regressor'], mdl['parents'],
                     mdl['body_dof'], mdl['hand_dof'], mdl['hands_mean'], mdl['selected_components'])
    pr = capi.Prior(case['prior']['means'], case['prior']['chols'], case['prior']['weights'])
    return dev, pr


@pytest.mark.parametrize('fingers', [False, True])
def test_stagei_matches_oracle(fingers):
    from moshpp_amd import capi
    from oracle import stagei_oracle as s1
    c = helpers.stagei_case() if not fingers else helpers.stagei_case(finger_markers=True, M=36, seed=2)
    dev, pr = _device(c)
    kw = helpers.stagei_kwargs(c, optimize_fingers=fingers)
    t0 = time.time()
    out = capi.stagei_solve_host(dev, pr, **kw)
    t_gpu = time.time() - t0
    t0 = time.time()
    ref = s1.stagei_solve(c['m'], c['faces'], c['prior'], 'smplh', c['frames'], c['vids'], c['mask'], c['m2b'], c['nb'],
                          optimize_fingers=fingers)
    t_cpu = time.time() - t0
    print(f'stagei fingers={fingers}: gpu {t_gpu:.3f} s ({out["iters"]} iterations), oracle {t_cpu:.3f} s')
    # tolerances: f64 on both sides; the GPU contracts to FMA and sums in a different order, the dogleg amplifies that a little
    assert np.abs(out['betas'] - ref['betas']).max() < 1e-5
    assert np.abs(out['markers_latent'] - ref['markers_latent']).max() < 1e-6      # metres
    assert np.abs(out['pose'] - ref['pose']).max() < 1e-5 and np.abs(out['trans'] - ref['trans']).max() < 1e-6
    assert (out['markers_latent_vids'] == ref['markers_latent_vids']).all()
    e = ref['errs']
    want = dict(data=e['data'], poseB=e['poseB'], init=e['init_0'], beta=e['beta'], surf=e['surf'], poseH=e.get('poseH', 0.0))
    for k, v in want.items():
        assert abs(out['errs'][k] - v) <= 1e-5 * max(1.0, abs(v)), k


def test_stagei_rejects_bad_input():
    from moshpp_amd import capi
    c = helpers.stagei_case()
    dev, pr = _device(c)
    kw = helpers.stagei_kwargs(c)
    kw['frames'] = [(np.array([0, 1, c['M'] + 3]), np.zeros((3, 3)))] + list(kw['frames'][1:])
    with pytest.raises(capi.MoshiiError):
        capi.stagei_solve_host(dev, pr, **kw)
