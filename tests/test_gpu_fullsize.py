"""BASELINE.json configurations at their FULL sizes on the GPU, checked through properties that need no full-size oracle
run: every frame solved, markers reproduced to the noise level, the chain is a deterministic function of its state
(continuation from a recorded state reproduces the rest bit for bit), the chunk-parallel solve equals the sequential
chain on every frame, plus the oracle itself on a leading stretch.  (The 4000-frame SMPL-H oracle comparison over ALL
frames lives in tools/full_parity.py / profiles/r01_full_parity.txt: 157 s of CPU.)"""
import numpy as np
import pytest

from oracle import stageii_oracle as so

pytestmark = pytest.mark.gpu
TIGHT = 1e-7


def _oracle_head(job, solver, n):
    """Oracle chain on the first n frames of a workload.make_job job, built from the solver's own attachment."""
    sm = job['sm']
    model = dict(v_template=sm.v_template, shapedirs=sm.shapedirs, posedirs=sm.posedirs, weights=sm.weights,
                 J_regressor=sm.J_regressor, parents=sm.parents, body_dof=sm.body_dof, hand_dof=sm.hand_dof,
                 hands_mean=sm.hands_mean, selected_components=sm.selected_components)
    m = so.prepare_model(model, solver.betas)
    return so.stageii_chain(m, job['prior'], solver.tc.closest.astype(np.int64), solver.tc.coef, job['obs'][:n],
                            job['vis'][:n], job['model_type'], optimize_fingers=job['optimize_fingers'])


def _check_properties(job, solver, out, noise_rmse, head=24):
    from moshpp_amd import capi
    F = job['vis'].shape[0]
    vis, obs = job['vis'], job['obs']
    has = vis.any(1)
    assert np.array_equal(out['status'] == 0, has) and np.all(out['status'][~has] == 1)
    # (1) simulated markers reproduce the observations to the noise level on every solved frame
    d = (out['markers_sim'] - obs)[vis]
    rmse = float(np.sqrt((d ** 2).sum(1).mean()))
    assert rmse < noise_rmse, rmse
    # (2) the oracle on the leading frames
    ref = _oracle_head(job, solver, head)
    solved = np.flatnonzero(has[:head])
    assert np.abs(out['fullpose'][solved] - ref['fullpose']).max() < TIGHT
    assert np.abs(out['trans'][solved] - ref['trans']).max() < TIGHT
    np.testing.assert_array_equal(out['iters'][solved, 0], ref['iters'])
    # (3) continuation: restart in the middle from the recorded state -> the rest of the chain, bit for bit
    t0 = F // 2 + 3
    prev = np.flatnonzero(has[:t0])
    tail = capi.chain_solve_host(solver.dev, solver.prior, solver.opts,
                                 [dict(attach=solver.attach, obs=obs[t0:], vis=vis[t0:], first=False,
                                       init_pose=out['pose'][prev[-1]], init_trans=out['trans'][prev[-1]],
                                       init_pose_prev=out['pose'][prev[-2]])])[0]
    assert np.array_equal(tail['fullpose'], out['fullpose'][t0:]) and np.array_equal(tail['trans'], out['trans'][t0:])
    return rmse


def test_config1_smpl_120_frames_41_markers_vs_oracle(gpu_lib):
    """BASELINE configs[0] in full: the whole 120-frame SMPL sequence against the oracle."""
    from moshpp_amd import workload
    job = workload.make_job('smpl', 120, 41, seed=70)
    solver = workload.make_solver(job)
    out = solver.solve(job['obs'], job['vis'])
    _check_properties(job, solver, out, 1.5e-3, head=120)


def test_config2_smplh_4000_frames_chunked_equals_sequential(gpu_lib):
    """BASELINE configs[1] in full (4000 frames, 53 markers): properties + chunk-parallel == sequential on all frames."""
    from moshpp_amd import workload
    job = workload.make_job('smplh', 4000, 53, seed=71)
    solver = workload.make_solver(job)
    seq = solver.solve(job['obs'], job['vis'])
    rmse = _check_properties(job, solver, seq, 1.5e-3)
    chk = solver.solve(job['obs'], job['vis'], chain_mode='chunked')
    rep = chk['chunk_report']
    dp = np.abs(chk['fullpose'] - seq['fullpose']).max()
    print(f'smplh 4000f: marker rmse {rmse:.2e} m; chunks {rep["n_chunks"]} repaired {rep["n_repaired"]} in '
          f'{rep["repair_rounds"]} rounds; max|chunked - sequential| {dp:.2e} rad')
    assert rep['n_chunks'] > 100 and rep['max_handoff_dev'] <= rep['verify_tol']
    assert dp < 1e-8 and np.abs(chk['trans'] - seq['trans']).max() < 1e-8
    assert np.array_equal(chk['status'], seq['status'])


@pytest.mark.parametrize('hand,M,seed', [('left', 34, 72), ('right', 33, 73)])
def test_config4_mano_10000_frames_per_hand(gpu_lib, hand, M, seed):
    """BASELINE configs[3] in full: a MANO hand with 34 / 33 markers over 10 000 frames (each hand is its own model and
    chain, as the reference runs them), per-frame dogleg with the hand-PCA coefficients free."""
    from moshpp_amd import workload
    job = workload.make_job('mano', 10000, M, seed=seed, optimize_fingers=True)
    solver = workload.make_solver(job)
    seq = solver.solve(job['obs'], job['vis'])
    rmse = _check_properties(job, solver, seq, 1.5e-3)
    chk = solver.solve(job['obs'], job['vis'], chain_mode='chunked')
    dp = np.abs(chk['fullpose'] - seq['fullpose']).max()
    print(f'mano {hand} 10000f: marker rmse {rmse:.2e} m, max|chunked - sequential| {dp:.2e} rad, '
          f'repairs {chk["chunk_report"]["n_repaired"]}/{chk["chunk_report"]["n_chunks"]}')
    assert dp < 1e-8


def test_config3_smplx_face_and_hands_many_sequences(gpu_lib):
    """BASELINE configs[2] shape at reduced length: 8 SMPL-X sequences, 89 markers incl. face / hand vertices, fingers +
    jaw + the yaml-default 80 expression coefficients free (194 unknowns), one launch; copies of one sequence must agree
    bit for bit, and the oracle bounds the first frames."""
    from moshpp_amd import capi
    from tests.helpers import shape_case, device_case
    F, E = 40, 80
    case = shape_case('smplx', F=F, M=89, E=E, seed=21, kind='expr')
    dev = device_case(case, optimize_fingers=True, optimize_face=True, shape_kind='expr')
    outs = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'],
                                 [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True) for _ in range(8)])
    for o in outs[1:]:
        assert np.array_equal(o['fullpose'], outs[0]['fullpose']) and np.array_equal(o['shape'], outs[0]['shape'])
    o = outs[0]
    assert np.all(o['status'] == 0)
    d = (o['markers_sim'] - case['obs'])[case['vis']]
    # (the expression regulariser, weight 1, biases the strongly boosted synthetic expression block towards 0: ~1 cm)
    assert np.sqrt((d ** 2).sum(1).mean()) < 2e-2
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'][:5], case['vis'][:5], 'smplx',
                           optimize_fingers=True, optimize_face=True, free_shape='expr')
    assert np.abs(o['fullpose'][:5] - ref['fullpose']).max() < 1e-6 and np.abs(o['shape'][:5] - ref['shape']).max() < 1e-5
