"""The parity criterion itself (tests/parity_envelope.py) and the committed oracle trajectories it reads (CPU only)."""
import numpy as np
import pytest

from tests import parity_envelope as pe

BENCH_SEEDS = (1000, 123, 71, 5, 2024, 7)


@pytest.mark.parametrize('seed', BENCH_SEEDS + (11,))
def test_committed_trajectory_is_whole_and_consistent_with_itself(seed):
    g = pe.load(seed)
    assert g['pose'].shape[0] == g['trans'].shape[0] == len(g['frame_ids']) == len(g['spread']) == len(g['iters']) == 4000
    assert np.isfinite(g['pose']).all() and np.isfinite(g['spread']).all() and g['spread'].min() >= 0
    rep = pe.check(seed, g['pose'], g['trans'], g['iters'])
    assert rep['frames_outside_tolerance'] == 0 and rep['frames_parted_on_a_knife_edge'] == 0 and rep['max_abs_pose_diff_rad'] == 0.0
    # most of every sequence is well conditioned: the envelope is not a blanket excuse
    assert rep['well_conditioned_frames'] >= 3700, rep      # (seed 123: 3736)


def test_the_fixture_is_the_oracle_of_this_tree():
    """The first frames of the seed-1000 trajectory, recomputed now by oracle/stageii_oracle.py (2 s): a changed oracle must not be
    checked against trajectories of an older one."""
    from moshpp_amd import workload
    from oracle import stageii_oracle as so
    job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=1000)
    sm = job['sm']
    m = so.prepare_model(dict(v_template=sm.v_template, shapedirs=sm.shapedirs, posedirs=sm.posedirs, weights=sm.weights,
                              J_regressor=sm.J_regressor, parents=sm.parents, body_dof=sm.body_dof, hand_dof=sm.hand_dof,
                              hands_mean=sm.hands_mean, selected_components=sm.selected_components), job['betas'])
    pr = so.prepare_gmm_prior(job['seq']['gmm'], 63)
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3))
    closest, coef = so.transformed_coeffs(can, job['markers_latent'])
    n = 40
    ref = so.stageii_chain(m, pr, closest, coef, job['obs'][:n], job['vis'][:n], 'smplh')
    g = pe.load(1000)
    k = len(ref['frame_ids'])
    assert np.array_equal(ref['frame_ids'], g['frame_ids'][:k])
    assert np.abs(ref['pose'] - g['pose'][:k]).max() < 1e-12 and np.array_equal(np.asarray(ref['iters']), g['iters'][:k])


def test_criterion_semantics():
    g = pe.load(123)                       # the seed with knife edges
    ill = pe.dilated(g['spread']) > pe.WELL
    assert 50 < ill.sum() < 600
    f_ill = int(np.flatnonzero(g['spread'] == g['spread'].max())[0])
    f_well = int(np.flatnonzero(~ill)[len(np.flatnonzero(~ill)) // 3])
    # (1) a deviation that begins on a well-conditioned frame is outside the tolerance, however small above TIGHT
    p = g['pose'].copy(); p[f_well:f_well + 7] += 3e-7
    r = pe.check(123, p, g['trans'])
    assert r['frames_outside_tolerance'] == 7 and r['first_frames_outside'][0] == g['frame_ids'][f_well]
    # (2) ... below TIGHT it is round-off
    p = g['pose'].copy(); p[f_well:f_well + 7] += 5e-8
    assert pe.check(123, p, g['trans'])['frames_outside_tolerance'] == 0
    # (3) a trajectory that parts ON a knife edge and stays away until it re-converges is counted as parted, not outside ...
    p = g['pose'].copy(); p[f_ill:f_ill + 300] += 2e-3
    r = pe.check(123, p, g['trans'])
    assert r['frames_outside_tolerance'] == 0 and r['frames_parted_on_a_knife_edge'] == 300 and r['frames_over_1e-4_rad'] == 300
    # (4) ... unless it leaves the ceiling
    p = g['pose'].copy(); p[f_ill:f_ill + 5] += 3.0
    assert pe.check(123, p, g['trans'])['frames_outside_tolerance'] == 5
    # (5) once re-converged, a new deviation on well-conditioned frames is outside again
    p = g['pose'].copy(); p[f_ill:f_ill + 30] += 2e-3
    far = f_ill + 30 + int(np.flatnonzero(~ill[f_ill + 30:])[40])
    p[far] += 1e-5
    assert pe.check(123, p, g['trans'])['frames_outside_tolerance'] == 1
