"""CPU tests of the oracle itself (not gpu): self-checks that stand in for the missing reference tests
(SURVEY.md 8c): finite differences for every analytic Jacobian block, Rodrigues vs SciPy, the dogleg
minimiser vs scipy.optimize.least_squares, ground-truth recovery, and the committed golden vectors."""
import os

import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation as Rot

from oracle import stageii_oracle as so
from tests.helpers import oracle_case

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'oracle_golden.npz')


@pytest.fixture(scope='module')
def smplh_case():
    return oracle_case('smplh', F=8, M=53, seed=12, empty_frames=(3,))


def test_rodrigues_matches_scipy_and_log_roundtrip():
    rng = np.random.default_rng(0)
    for scale in (1e-9, 1e-4, 1e-2, 1.0, 3.0):
        r = rng.normal(0, 1, 3)
        r = r / np.linalg.norm(r) * scale
        R, Jl = so.rodrigues(r)
        assert np.abs(R - Rot.from_rotvec(r).as_matrix()).max() < 1e-14
        if scale > 1e-3:
            assert np.abs(so.rotmat_to_rotvec(R) - r).max() < 1e-10
        # dR/dr_c = [Jl[:,c]]x R  vs central differences
        eps = 1e-6
        for c in range(3):
            d = np.zeros(3); d[c] = eps
            num = (so.rodrigues(r + d)[0] - so.rodrigues(r - d)[0]) / (2 * eps)
            ana = so._skew(Jl[:, c]).dot(R)
            assert np.abs(num - ana).max() < 1e-8


def test_vertex_jacobian_finite_differences(smplh_case):
    m = smplh_case['m']
    rng = np.random.default_rng(1)
    vids = np.array([5, 100, 3000, 6000, 42])
    for fp in (so.fullpose_from_pose(m, rng.normal(0, 0.3, m['NP'])), np.r_[np.zeros(3), 1e-5 * np.ones(3), np.zeros(m['P'] - 6)]):
        trans = rng.normal(0, 1, 3)
        v, dv = so.verts_jacobian(m, fp, trans, vids)
        assert np.abs(v - so.verts_forward(m, fp, trans, vids)).max() < 1e-14
        eps = 1e-6
        for d in range(m['P']):
            a = fp.copy(); a[d] += eps
            b = fp.copy(); b[d] -= eps
            num = (so.verts_forward(m, a, trans, vids) - so.verts_forward(m, b, trans, vids)) / (2 * eps)
            assert np.abs(num - dv[:, :, d]).max() < 2e-9


def test_objective_jacobian_finite_differences(smplh_case):
    c = smplh_case
    m = c['m']
    rng = np.random.default_rng(2)
    root, body, finger, st1, st2 = so.pose_id_sets('smplh', m['NP'], True)
    o = so.StageIIObjective(m, c['closest'], c['coef'], c['prior'], body)
    o.pose = rng.normal(0, 0.1, m['NP']); o.trans = rng.normal(0, 1, 3)
    o.vis = c['vis'][4]; o.obs = c['obs'][4]; o.wt_data = 300.; o.wt_pose = 1.6
    o.velo_target = o.pose * 0.9; o.wt_velo = 2.5
    o.finger_ids = np.array(finger); o.wt_poseH = 1.0; o.free_ids = st2
    x = o.x()
    J = o.J(x)
    eps = 1e-6
    for d in range(0, len(x), 7):
        a = x.copy(); a[d] += eps
        b = x.copy(); b[d] -= eps
        num = (o.r(a) - o.r(b)) / (2 * eps)
        assert np.abs(num - J[:, d]).max() < 5e-6 * max(1.0, np.abs(J[:, d]).max())


def test_dogleg_reaches_scipy_minimum(smplh_case):
    c = smplh_case
    m = c['m']
    root, body, finger, st1, st2 = so.pose_id_sets('smplh', m['NP'], False)
    o = so.StageIIObjective(m, c['closest'], c['coef'], None, [])   # smooth objective (no max-mixture switch)
    o.vis = c['vis'][0]; o.obs = c['obs'][0]; o.wt_data = 400.; o.free_ids = st1
    o.velo_target = np.zeros(m['NP']); o.wt_velo = 2.5
    sim = o.markers_sim()
    R, T = so.rigid_landmark_transform(sim[o.vis].T, o.obs[o.vis].T)
    o.pose[:3] = so.rotmat_to_rotvec(R); o.trans[:] = T.ravel()
    x0 = o.x()
    xd = so.minimize_dogleg(o, x0, e_3=0.0, delta_0=.5, maxiter=60)
    ls = least_squares(o.r, x0, jac=o.J, method='trf', xtol=1e-14, ftol=1e-14, gtol=1e-12)
    cost_d = np.sum(o.r(xd) ** 2)
    assert cost_d <= 2 * ls.cost * (1 + 1e-8)
    assert np.abs(xd - ls.x).max() < 1e-5


def test_ground_truth_recovery_and_skip_rule(smplh_case):
    c = smplh_case
    res = so.stageii_chain(c['m'], c['prior'], c['closest'], c['coef'], c['obs'], c['vis'], 'smplh')
    assert 3 not in res['frame_ids'] and len(res['frame_ids']) == 7      # chmosh.py:586-588
    assert 'velo' in res['errs'] and len(res['errs']['velo']) == len(res['frame_ids']) - 2   # :624-626,656-657
    gt = c['s']['trans_gt'][res['frame_ids']]
    assert np.abs(res['trans'] - gt).max() < 5e-3
    sq = [((a - c['obs'][t][c['vis'][t]]) ** 2).sum(1) for a, t in zip(res['markers_sim'], res['frame_ids'])]
    assert np.sqrt(np.concatenate(sq).mean()) < 2e-3   # noise is 0.5 mm per axis


@pytest.mark.parametrize('name,mt,F,M,seed,fingers', [('smpl_41mk_10f', 'smpl', 10, 41, 11, False),
                                                      ('smplh_53mk_8f', 'smplh', 8, 53, 12, False),
                                                      ('mano_24mk_8f', 'mano', 8, 24, 13, True)])
def test_oracle_matches_golden(name, mt, F, M, seed, fingers):
    g = np.load(GOLDEN)
    case = oracle_case(mt, F=F, M=M, seed=seed, empty_frames=(3,))
    np.testing.assert_allclose(np.array([case['obs'].sum(), case['vis'].sum(), case['coef'].sum()]),
                               g[f'{name}/obs_checksum'], rtol=1e-12)
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'], mt,
                           optimize_fingers=fingers)
    np.testing.assert_array_equal(ref['frame_ids'], g[f'{name}/frame_ids'])
    np.testing.assert_array_equal(ref['iters'], g[f'{name}/iters'])
    assert np.abs(ref['fullpose'] - g[f'{name}/fullpose']).max() < 1e-9
    assert np.abs(ref['trans'] - g[f'{name}/trans']).max() < 1e-9


def test_reference_cost_mode_is_the_same_arithmetic(smplh_case):
    c = smplh_case
    a = so.stageii_chain(c['m'], c['prior'], c['closest'], c['coef'], c['obs'][:2], c['vis'][:2], 'smplh')
    b = so.stageii_chain(c['m'], c['prior'], c['closest'], c['coef'], c['obs'][:2], c['vis'][:2], 'smplh',
                         reference_cost=True)
    assert np.abs(a['fullpose'] - b['fullpose']).max() < 1e-10


@pytest.mark.parametrize('model_type,kind', [('smplx', 'expr'), ('smplh', 'dmpl')])
def test_free_shape_block_jacobian_and_chain(model_type, kind):
    """Expression / DMPL coefficients as Step-2 free variables (chmosh.py:685-699): analytic Jacobian of the full
    objective incl. the shape columns (the restated `lbs_derivatives_wrt_shape`) vs central differences, and the chain
    recovers the time-varying coefficients it was generated with."""
    from tests.helpers import shape_case
    c = shape_case(model_type, F=5, M=36, E=5, seed=2, kind=kind)
    m = c['m']
    face = kind == 'expr'
    root, body, finger, st1, st2 = so.pose_id_sets(model_type, m['NP'], False, False, optimize_face=face)
    o = so.StageIIObjective(m, c['closest'], c['coef'], c['prior'], body)
    rng = np.random.default_rng(4)
    o.pose = c['pose_gt'][2] + rng.normal(0, 0.02, m['NP'])
    o.trans = c['s']['trans_gt'][2].copy()
    o.shp = rng.normal(0, 0.5, c['E'])
    o.vis, o.obs = c['vis'][2], c['obs'][2]
    o.wt_data, o.wt_pose, o.wt_velo = 400 * 46 / o.vis.sum(), 1.6, 2.5
    o.velo_target = o.pose + rng.normal(0, 0.01, m['NP'])
    o.free_ids = st2
    if face:
        assert st2[-3:] == [66, 67, 68] or set([66, 67, 68]) <= set(st2)
        o.face_ids, o.wt_poseF = np.array([66, 67, 68]), 1.3
    o.shape_free, o.wt_shape = True, 0.9
    o.shp_anchor, o.wt_stay = o.shp + 0.1, (6.0 if kind == 'dmpl' else 0.0)
    x = o.x()
    assert len(x) == 3 + len(st2) + c['E']
    J = o.J(x)
    r0 = o.r(x)
    assert J.shape == (len(r0), len(x))
    num = np.zeros_like(J)
    for q in range(len(x)):
        h = 1e-6
        a, b = x.copy(), x.copy()
        a[q] += h
        b[q] -= h
        num[:, q] = (o.r(a) - o.r(b)) / (2 * h)
    assert np.abs(num - J).max() < 2e-5 * max(1.0, np.abs(J).max())
    assert np.abs(J[:, -c['E']:]).max() > 1.0           # the shape columns are live
    res = so.stageii_chain(m, c['prior'], c['closest'], c['coef'], c['obs'], c['vis'], model_type,
                           optimize_face=face, free_shape=kind)
    assert res['shape'].shape == (5, c['E'])
    err = np.abs(res['shape'] - c['shp_gt'])
    assert err[1:].max() < 0.35, err.max()             # regularised (wt 1.0) -> biased towards 0, but tracks
    rm = np.sqrt(np.mean([np.mean((a - c['obs'][t][c['vis'][t]]) ** 2) for t, a in enumerate(res['markers_sim'])]))
    assert rm < 1.5e-3
    assert ('shape_stay' in res['errs']) == (kind == 'dmpl')
    if kind == 'dmpl':
        assert len(res['errs']['shape_stay']) == 4      # from the second solved frame on
    if face:
        assert np.abs(res['pose'][:, 66:69]).max() > 1e-3   # the jaw is free (weakly observed by this layout)


@pytest.mark.parametrize('name,mt,kind,E,F,M,seed', [('smplx_expr5_6f', 'smplx', 'expr', 5, 6, 40, 31),
                                                       ('smplh_dmpl4_6f', 'smplh', 'dmpl', 4, 6, 40, 32)])
def test_oracle_reproduces_committed_shape_goldens(name, mt, kind, E, F, M, seed):
    """The Step-2 extras (jaw + expression, DMPL) frozen in tests/golden/oracle_golden.npz: later edits of the oracle or of the
    synthetic generators cannot silently move this parity target either."""
    from tests.helpers import shape_case
    g = np.load(GOLDEN)
    case = shape_case(mt, F=F, M=M, E=E, seed=seed, kind=kind)
    assert np.allclose(g[f'{name}/obs_checksum'], [case['obs'].sum(), case['vis'].sum(), case['coef'].sum()], rtol=1e-12)
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'], mt,
                           optimize_face=kind == 'expr', free_shape=kind)
    assert np.abs(ref['fullpose'] - g[f'{name}/fullpose']).max() < 1e-9
    assert np.abs(ref['shape'] - g[f'{name}/shape']).max() < 1e-9
    np.testing.assert_array_equal(ref['iters'], g[f'{name}/iters'])
