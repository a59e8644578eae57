"""GPU parity: libmoshii (HIP, through the C ABI) against the float64 oracle on the same seeded inputs.
Tolerances (BASELINE.json north_star): markers 1e-3 m RMSE, pose 1e-4 rad.  The HIP path is float64
with the oracle's formulas, so the tests below hold it to far tighter bounds and state them."""
import numpy as np
import pytest

from oracle import stageii_oracle as so
from tests.helpers import oracle_case, device_case

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4      # rad, north_star
MARKER_TOL = 1e-3    # m RMSE, north_star
TIGHT = 1e-7         # what two float64 implementations of the same formulas actually deliver


def test_lbs_f64_matches_oracle(gpu_lib):
    case = oracle_case('smplh', F=4, M=53, seed=3)
    dev = device_case(case)
    m = case['m']
    rng = np.random.default_rng(0)
    pose = rng.normal(0, 0.3, (3, m['NP']))
    trans = rng.normal(0, 1, (3, 3))
    got = dev['model'].lbs_forward(pose, trans)
    for f in range(3):
        ref = so.verts_forward(m, so.fullpose_from_pose(m, pose[f]), trans[f])
        assert np.abs(got[f] - ref).max() < 1e-12
    got32 = dev['model'].lbs_forward(pose, trans, dtype=np.float32)
    ref = np.stack([so.verts_forward(m, so.fullpose_from_pose(m, pose[f]), trans[f]) for f in range(3)])
    assert np.abs(got32 - ref).max() < 5e-6   # float32 export kernel: micrometres
    np.testing.assert_allclose(dev['model'].joints(), m['J'], atol=1e-13)


def test_attach_markers_matches_oracle(gpu_lib):
    case = oracle_case('smplh', F=4, M=53, seed=4)
    dev = device_case(case)
    m = case['m']
    rng = np.random.default_rng(1)
    pose = rng.normal(0, 0.3, (5, m['NP']))
    trans = rng.normal(0, 1, (5, 3))
    got = dev['attach'].markers(pose, trans)
    o = so.StageIIObjective(m, case['closest'], case['coef'], case['prior'], [])
    for f in range(5):
        ref = o.markers_sim(pose[f], trans[f])
        assert np.abs(got[f] - ref).max() < 1e-12


@pytest.mark.parametrize('model_type,M,fingers,seed', [
    ('smplh', 53, False, 0),
    ('smpl', 41, False, 1),
    ('smplh', 53, True, 2),
    ('mano', 24, True, 5),
    ('smplx', 60, False, 6),
])
def test_chain_parity(gpu_lib, model_type, M, fingers, seed):
    from moshpp_amd import capi
    F = 24
    kw = dict(body_only_markers=not fingers) if model_type != 'mano' else {}
    case = oracle_case(model_type, F=F, M=M, seed=seed, empty_frames=(7,), **kw)
    dev = device_case(case, optimize_fingers=fingers)
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'],
                           model_type, optimize_fingers=fingers)
    out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'],
                                [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
    solved = np.where(out['status'] == 0)[0]
    assert list(solved) == list(ref['frame_ids'])            # frame 7 has no markers and is skipped
    assert out['status'][7] == 1
    dp = np.abs(out['fullpose'][solved] - ref['fullpose']).max()
    dt = np.abs(out['trans'][solved] - ref['trans']).max()
    sq = []
    for i, t in enumerate(solved):
        vm = case['vis'][t]
        sq.append(((out['markers_sim'][t][vm] - ref['markers_sim'][i]) ** 2).sum(1))
    rmse = np.sqrt(np.concatenate(sq).mean())
    print(f'{model_type}: max|dpose|={dp:.3e} rad max|dtrans|={dt:.3e} m marker rmse={rmse:.3e} m '
          f'iters gpu={out["iters"][solved, 0].sum()} oracle={ref["iters"].sum()}')
    assert dp < POSE_TOL and rmse < MARKER_TOL and dt < 1e-4
    assert dp < TIGHT and rmse < TIGHT, 'float64 HIP path drifted from the oracle beyond round-off'
    np.testing.assert_array_equal(out['iters'][solved, 0], ref['iters'])
    np.testing.assert_allclose(out['errs'][solved, 0], ref['errs']['data'], rtol=1e-6)
    if 'poseB' in ref['errs']:
        np.testing.assert_allclose(out['errs'][solved, 1], ref['errs']['poseB'], rtol=1e-6)
    if 'velo' in ref['errs']:
        np.testing.assert_allclose(out['errs'][solved[2:], 2], ref['errs']['velo'], rtol=1e-6, atol=1e-12)
