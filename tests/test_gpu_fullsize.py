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
    rmse = _check_properties(job, solver, seq, 1.5e-3, head=400)     # the oracle on the first 400 frames (10 s of CPU)
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


@pytest.mark.parametrize('kind,model_type', [('expr', 'smplx'), ('dmpl', 'smplh')])
def test_sequence_solve_carries_the_free_shape_block_on_the_gpu(gpu_lib, kind, model_type):
    """moshii_sequence_solve with n_shape > 0 on the device (round 1 had this in emulation only): the expression / DMPL
    coefficients travel in the chunk hand-off states; the stitched result equals the sequential chain."""
    from moshpp_amd import capi
    from tests.helpers import shape_case, device_case
    F = 96
    case = shape_case(model_type, F=F, M=40, E=4, seed=9, kind=kind)
    dev = device_case(case, optimize_face=(kind == 'expr'), shape_kind=kind)
    seq = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'],
                                [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
    outs, rep = capi.sequence_solve_host(dev['model'], dev['prior'], dev['opts'],
                                         [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'])],
                                         num_chunks=6, warmup=8, verify_tol=1e-9)
    print(f'{kind}: {rep}')
    assert rep['n_chunks'] == 6 and np.abs(seq['shape']).max() > 0.2
    assert np.abs(outs[0]['fullpose'] - seq['fullpose']).max() < 1e-7 and np.abs(outs[0]['shape'] - seq['shape']).max() < 1e-7
    assert np.array_equal(outs[0]['status'], seq['status'])


OVER_TOL_BUDGET = 8      # frames of a 4000-frame sequence and mode that may differ from the oracle by more than the north-star 1e-4 rad
                         # (all of them parted on a knife edge the oracle's own perturbed runs open; measured: seed 123, 3-4 per mode)


def _envelope_report(case, out, vis, oracle_model, with_iters=True, shape=False):
    """pe.check of a device result dict over the frames it solved, simulated markers included."""
    from tests import parity_envelope as pe
    solved = np.flatnonzero(out['status'] <= 0) if out['status'].min() < 0 else np.flatnonzero(out['status'] == 0)
    fid = pe.load(case)['frame_ids']
    solved = solved[solved <= fid[-1]]
    return pe.check(case, out['pose'][solved], out['trans'][solved], out['iters'][solved] if with_iters else None, frames=solved,
                    shape=out['shape'][solved] if shape else None, markers_sim=out['markers_sim'][solved], vis=vis[solved], oracle_model=oracle_model)


@pytest.mark.parametrize('seed', [1000, 123, 71, 5, 2024, 7])
def test_both_modes_lie_inside_the_oracle_envelope_on_every_frame_of_every_bench_seed(gpu_lib, seed):
    """All six bench sequences, all 4000 frames, BOTH modes AS SHIPPED -- `chain_mode='sequential'` (the cooperative chain) and the
    drop-in default (`chain_mode='auto'` = verified chunks for this workload, at the default hand-off tolerance 1e-9) -- against the
    committed oracle trajectory of the seed and its sensitivity envelope (tests/parity_envelope.py: 1e-7 rad, equal dogleg iteration
    counts and 1e-6 m marker RMSE per frame wherever K perturbed oracle runs stay within 3e-9 rad of each other; where the reference's
    own algorithm sits on a knife edge a trajectory may part by max(1e-3, 30 x the stretch's spread) and has to be back within 64
    frames; whole-sequence marker RMSE against the ORACLE's simulated markers <= 1e-3 m).  No frame of any sequence is exempted by
    name; a frame outside the criterion fails the test.  The chunked scheme has timing-dependent paths: three runs."""
    from moshpp_amd import workload
    from tests import parity_envelope as pe
    from tests.helpers import oracle_of_job
    job = workload.make_job('smplh', 4000, 53, seed=seed)
    solver = workload.make_solver(job)
    m, _, closest, coef = oracle_of_job(job)
    om = (m, closest, coef)
    assert np.array_equal(closest, solver.tc.closest) and np.abs(coef - solver.tc.coef).max() < 1e-12
    seq = solver.solve(job['obs'], job['vis'], chain_mode='sequential')
    rep = _envelope_report(seed, seq, job['vis'], om)
    print(f'seed {seed} sequential vs oracle: {rep}')
    assert rep['frames'] == len(pe.load(seed)['frame_ids']) and pe.ok(rep), rep
    assert rep['frames_over_1e-4_rad'] <= OVER_TOL_BUDGET, rep
    assert solver.choose_chain_mode(4000) == 'chunked'
    for _ in range(3):
        chk = solver.solve(job['obs'], job['vis'], chain_mode='auto')   # what mosh_stageii ships: auto -> chunked, verify_tol 1e-9
        assert chk['chunk_report']['verify_tol'] == 1e-9 and np.array_equal(chk['status'], seq['status'])
        rc = _envelope_report(seed, chk, job['vis'], om, with_iters=False)
        rs = pe.compare(seed, chk, seq)
        print(f'seed {seed} default (chunked) vs oracle: outside {rc["frames_outside_tolerance"]}, well {rc["max_dev_on_well_conditioned_frames_rad"]:.1e}, '
              f'parted {rc["frames_parted_on_a_knife_edge"]} frames (max {rc["max_dev_on_parted_frames_rad"]:.1e} rad), marker rmse vs oracle {rc["marker_rmse_vs_oracle_m"]:.1e} m '
              f'(worst frame {rc["worst_frame_marker_rmse_vs_oracle_m"]:.1e}); vs sequential: outside {rs["frames_outside_tolerance"]}; {chk["chunk_report"]["n_repaired"]} chunks repaired')
        assert pe.ok(rc) and rs['frames_outside_tolerance'] == 0, (rc, rs)
        assert rc['frames_over_1e-4_rad'] <= OVER_TOL_BUDGET, rc


@pytest.mark.parametrize('case,hand_markers', [('mano_72', 34), ('mano_73', 33)])
def test_config4_mano_every_frame_against_the_oracle(gpu_lib, case, hand_markers):
    """BASELINE configs[3] at its stated length: all 10 000 frames of each MANO hand (hand-PCA coefficients free, no pose prior) against
    the committed oracle trajectory + envelope (tests/golden/make_oracle_trajectories_configs.py) -- the sequential chain (iteration
    counts too) and the chunked solve, simulated markers against the oracle's."""
    from moshpp_amd import workload
    from tests import parity_envelope as pe
    from tests.helpers import oracle_of_job
    seed = int(case.split('_')[1])
    job = workload.make_job('mano', 10000, hand_markers, seed=seed, optimize_fingers=True)
    solver = workload.make_solver(job)
    m, _, closest, coef = oracle_of_job(job)
    assert np.array_equal(closest, solver.tc.closest) and np.abs(coef - solver.tc.coef).max() < 1e-12
    for mode in ('sequential', 'chunked'):
        out = solver.solve(job['obs'], job['vis'], chain_mode=mode)
        rep = _envelope_report(case, out, job['vis'], (m, closest, coef), with_iters=(mode == 'sequential'))
        print(f'{case} {mode} vs oracle over {rep["frames"]} frames: {rep}')
        assert rep['frames'] == 10000 and pe.ok(rep), rep
        assert rep['frames_over_1e-4_rad'] == 0, rep


def test_config5_first_8000_frames_against_the_oracle(gpu_lib):
    """BASELINE configs[4]'s Stage-II leg: the 50 000-frame SMPL-H capture solved as shipped (auto -> verified chunks); its first 8000
    frames against the committed oracle trajectory + envelope, simulated markers against the oracle's."""
    from moshpp_amd import workload
    from tests import parity_envelope as pe
    from tests.helpers import oracle_of_job
    job = workload.make_job('smplh', 50000, 53, seed=1000)
    solver = workload.make_solver(job)
    m, _, closest, coef = oracle_of_job(job)
    out = solver.solve(job['obs'], job['vis'], chain_mode='auto')
    assert out['chain_mode'] == 'chunked' and out['chunk_report']['n_chunks'] > 100
    rep = _envelope_report('config5_1000', out, job['vis'], (m, closest, coef), with_iters=False)
    print(f'config 5, first 8000 of 50000 frames (chunked) vs oracle: {rep}')
    assert rep['frames'] == 8000 and pe.ok(rep), rep
    assert rep['frames_over_1e-4_rad'] <= OVER_TOL_BUDGET, rep
    seq = solver.solve(job['obs'][:8000], job['vis'][:8000], chain_mode='sequential')
    rs = _envelope_report('config5_1000', seq, job['vis'][:8000], (m, closest, coef))
    print(f'config 5, first 8000 frames (sequential chain) vs oracle: {rs}')
    assert pe.ok(rs) and rs['frames_over_1e-4_rad'] <= OVER_TOL_BUDGET, rs


@pytest.mark.parametrize('case', ['config3_7000', 'config3_7001'])
def test_config3_capture_every_frame_against_the_oracle(gpu_lib, case):
    """BASELINE configs[2]: the 4000-frame captures 7000 and 7001 of the config-3 subject (SMPL-X, 89 markers, fingers + jaw + 80 expression
    coefficients free: 194 unknowns) as the library runs it (cooperative chain) and as one workgroup, EVERY frame against the committed
    oracle trajectory + envelope -- pose variables, translation, expression coefficients, dogleg iteration counts, simulated markers.
    (Round 4's profiles/r04_config3_full_parity.txt: the device parts from the oracle at frame ~2380 by 0.39 rad.  The oracle's own runs
    on observations perturbed by 1e-13 m part there too -- by up to 0.68 rad: the fingers + face block loses track for a stretch in every
    float64 execution -- which is what the envelope records and this test holds the device to.)"""
    from moshpp_amd import capi, workload
    from tests import parity_envelope as pe
    from tests.golden.make_oracle_trajectories_configs import case_inputs
    c = case_inputs(case)
    solver = workload.make_solver(c['job'])
    assert np.array_equal(c['closest'], solver.tc.closest) and np.abs(c['coef'] - solver.tc.coef).max() < 1e-12
    ch = [dict(attach=solver.attach, obs=c['obs'], vis=c['vis'], first=True)]
    for coop in (0, 1):
        o = capi.chain_solve_host(solver.dev, solver.prior, solver.opts, ch, coop=coop)[0]
        assert np.all(o['status'] <= 0) and (o['status'] != 0).mean() < 2e-3
        rep = _envelope_report(case, o, c['vis'], (c['m'], c['closest'], c['coef']), shape=True)
        print(f'config 3 {case} ({capi.last_launch_info()[0]}) vs oracle over {rep["frames"]} frames: {rep}')
        assert rep['frames'] == 4000 and pe.ok(rep), rep


def test_config5_50000_frames_one_sequence(gpu_lib):
    """BASELINE configs[4]'s Stage-II leg at size: one 50 000-frame SMPL-H capture, chunk-parallel; properties on every frame,
    the oracle on the leading frames, the sequential chain on a 3 000-frame window restarted from the recorded state."""
    from moshpp_amd import capi, workload
    F = 50000
    job = workload.make_job('smplh', F, 53, seed=1000)
    solver = workload.make_solver(job)
    out = solver.solve(job['obs'], job['vis'], chain_mode='chunked')
    rep = out['chunk_report']
    vis, obs = job['vis'], job['obs']
    has = vis.any(1)
    assert np.array_equal(out['status'] == 0, has)
    d = (out['markers_sim'] - obs)[vis]
    rmse = float(np.sqrt((d ** 2).sum(1).mean()))
    assert rmse < 1.5e-3 and rep['max_handoff_dev'] <= rep['verify_tol']
    ref = _oracle_head(job, solver, 60)
    solved = np.flatnonzero(has[:60])
    assert np.abs(out['fullpose'][solved] - ref['fullpose']).max() < TIGHT
    t0 = 31000
    prev = np.flatnonzero(has[:t0])
    win = capi.chain_solve_host(solver.dev, solver.prior, solver.opts,
                                [dict(attach=solver.attach, obs=obs[t0:t0 + 3000], vis=vis[t0:t0 + 3000], first=False,
                                      init_pose=out['pose'][prev[-1]], init_trans=out['trans'][prev[-1]],
                                      init_pose_prev=out['pose'][prev[-2]])])[0]
    dw = np.abs(win['fullpose'] - out['fullpose'][t0:t0 + 3000]).max()
    print(f'50000 frames: marker rmse {rmse:.2e} m, {rep}, max |window chain - chunked| {dw:.2e} rad')
    assert dw < 1e-7


def _config3_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', 'config3_oracle.npz'))


def _hold_to_config3_golden(g, ms, out, n_frames):
    """A chain's first `n_frames` frames of capture `ms` against the committed oracle trajectory (tests/golden/make_config3_golden.py):
    every frame's dogleg iteration count, every 10th frame's fullpose / expression / translation."""
    st = int(g['stride'])
    sel = np.arange(0, n_frames, st)
    k = len(sel)
    dp = np.abs(out['fullpose'][sel] - g[f'fullpose_{ms}'][:k]).max()
    ds = np.abs(out['shape'][sel] - g[f'shape_{ms}'][:k]).max()
    dt = np.abs(out['trans'][sel] - g[f'trans_{ms}'][:k]).max()
    same = out['iters'][:n_frames, 0] == g[f'iters_{ms}'][:n_frames]
    return dp, ds, dt, same


def test_config3_smplx_32_sequences_of_4000_frames(gpu_lib):
    """BASELINE configs[2] at its stated size: 32 SMPL-X sequences x 4000 frames, 89 markers incl. face / hand vertices, fingers +
    jaw + the yaml-default 80 expression coefficients free (194 unknowns per Step-2 solve), one launch in the reference's frame order --
    each chain a cooperative chain of 8 workgroups where the chip has 256 CUs (with a free expression block a chunk start never reproduces
    the chain's coefficients, DESIGN.md section 4a, so this size class runs sequentially per sequence).  The subject and capture are the
    bench's config-3 leg's (workload.make_face_job; capture 7000, generated on the host so that the committed oracle trajectory applies):
    copies agree bit for bit; every frame reproduces its markers; the oracle holds the 400-frame capture of the same subject and seed,
    solved in the same launch shape (every iteration count, every 10th frame's state)."""
    import time
    from moshpp_amd import capi, workload
    from tests.helpers import face_capture_host, face_job_oracle
    F, NSEQ, H = 4000, 32, 400
    job = workload.make_face_job()
    solver = workload.make_solver(job)
    m, pr, closest, coef = face_job_oracle(job)
    assert np.array_equal(closest, solver.tc.closest) and np.abs(coef - solver.tc.coef).max() < 1e-12    # the same attachment on both sides
    cap = face_capture_host(job, m, closest, coef, 7000, F)
    t0 = time.perf_counter()
    outs = capi.chain_solve_host(solver.dev, solver.prior, solver.opts,
                                 [dict(attach=solver.attach, obs=cap['obs'], vis=cap['vis'], first=True) for _ in range(NSEQ)])
    dt = time.perf_counter() - t0
    kernel = capi.last_launch_info()[0]
    for o in outs[1:]:
        assert np.array_equal(o['fullpose'], outs[0]['fullpose']) and np.array_equal(o['shape'], outs[0]['shape'])
    o = outs[0]
    # (every frame is solved; over 4000 frames the chain still meets a handful of ~10-frame stretches where the fingers + face block
    #  loses track -- data SSE in the hundreds, in the oracle as on the GPU, profiles/r04_config3_full_parity.txt -- and a frame of one
    #  of them can meet a non-positive pivot: flagged -1, solved by the Cauchy step)
    assert np.all(o['status'] <= 0) and (o['status'] != 0).mean() < 2e-3
    d = (o['markers_sim'] - cap['obs'])[cap['vis']]
    rmse = float(np.sqrt((d ** 2).sum(1).mean()))
    assert rmse < 5e-3          # (noise 0.5 mm; the expression regulariser holds the block towards 0)
    # the oracle's window: the committed trajectory is that of the capture GENERATED with 400 frames (the generator's noise / dropout
    # streams and its motion depend on the length: the 4000-frame capture's first 400 frames are another capture) -- the same launch
    # shape, 32 copies on 32 x 8 workgroups
    cap_h = face_capture_host(job, m, closest, coef, 7000, H)
    outs_h = capi.chain_solve_host(solver.dev, solver.prior, solver.opts,
                                   [dict(attach=solver.attach, obs=cap_h['obs'], vis=cap_h['vis'], first=True) for _ in range(NSEQ)])
    assert capi.last_launch_info()[0] == kernel
    for oh in outs_h[1:]:
        assert np.array_equal(oh['fullpose'], outs_h[0]['fullpose']) and np.array_equal(oh['shape'], outs_h[0]['shape'])
    g = _config3_golden()
    dp, ds, dtr, same = _hold_to_config3_golden(g, 7000, outs_h[0], H)
    print(f'config 3: {NSEQ} x {F} frames in {dt:.1f} s = {NSEQ * F / dt:.0f} frames/s ({kernel}); marker rmse {rmse:.2e} m; data SSE max {o["errs"][:, 0].max():.1f}; '
          f'{H}-frame capture vs oracle {dp:.2e} rad / {ds:.2e} (expression) / {dtr:.2e} m, iteration counts equal on {same.mean() * 100:.1f} % of frames')
    assert dp < 1e-6 and ds < 1e-6 and dtr < 1e-6
    assert same.all()


@pytest.mark.gpu
def test_an_ill_conditioned_stretch_is_the_algorithms_own_sensitivity(gpu_lib):
    """Seed 11 has a stretch of several hundred frames on which the chunked result departs from the sequential chain by 5e-2 rad at
    every hand-off tolerance.  That is not the chunk scheme: the sequential chain continued from its OWN state a few frames ahead of
    the stretch reproduces itself bit for bit, and continued from that state plus 1e-13 it takes the other side of a knife-edge dogleg
    decision and tracks another local solution -- whose simulated markers are 0.2 mm RMS from the first one's over the stretch
    (tools/knife_edge.py).  WHERE the stretch lies is read off the oracle's own sensitivity envelope
    (tests/golden/oracle_traj_seed11.npz: the longest run of frames on which perturbed oracle runs part), not written down here."""
    from moshpp_amd import capi, workload
    from tests import parity_envelope as pe
    ill = pe.dilated(pe.load(11)['spread']) > pe.WELL
    edges = np.flatnonzero(np.diff(np.concatenate([[0], ill.astype(int), [0]])))
    starts, ends = edges[0::2], edges[1::2]
    k = int(np.argmax(ends - starts))
    t0, t1 = int(starts[k]), int(min(ends[k] + 20, 4000))          # (the dilation puts t0 two dozen frames ahead of the first parted frame)
    assert t1 - t0 > 50 and t0 > 2
    job = workload.make_job('smplh', 4000, 53, seed=11)
    solver = workload.make_solver(job)
    seq = solver.solve(job['obs'][:t1], job['vis'][:t1])
    rng = np.random.default_rng(0)

    def continued(d):
        return capi.chain_solve_host(solver.dev, solver.prior, solver.opts,
                                     [dict(attach=solver.attach, obs=job['obs'][t0:t1], vis=job['vis'][t0:t1], first=False,
                                           init_pose=seq['pose'][t0 - 1] + d * rng.standard_normal(seq['pose'].shape[1]),
                                           init_trans=seq['trans'][t0 - 1], init_pose_prev=seq['pose'][t0 - 2])])[0]
    same = continued(0.0)
    assert np.array_equal(same['fullpose'], seq['fullpose'][t0:t1]) and np.array_equal(same['iters'], seq['iters'][t0:t1])
    other = continued(1e-13)
    dev = np.abs(other['fullpose'] - seq['fullpose'][t0:t1]).max(1)
    print(f'1e-13 perturbation at frame {t0}: max {dev.max():.2e} rad, first frame > 1e-7: {t0 + int(np.flatnonzero(dev > 1e-7)[0])}')
    assert dev[:10].max() < 1e-9 and dev.max() > 1e-3                        # tiny for a while, then amplified by orders of magnitude
    d2 = ((other['markers_sim'] - seq['markers_sim'][t0:t1]) ** 2).sum(-1)
    # ... into another solution of the same frames: the simulated markers of the two are 0.2 mm RMS apart over the stretch
    # (2.4 mm on its worst frame)
    assert np.sqrt(d2.mean()) < 1e-3 and np.sqrt(d2.mean(1)).max() < 5e-3


@pytest.mark.gpu
def test_chunked_vs_sequential_on_a_seed_with_a_long_ill_conditioned_stretch(gpu_lib):
    """Seed 11 (not a bench seed): the sequence whose knife edge the test above takes apart -- a stretch of several hundred frames on
    which two runs that differ by a hand-off tolerance follow different local solutions.  Same criterion as for the bench seeds
    (tests/parity_envelope.py over the committed oracle trajectory + envelope of seed 11): round-off outside the stretch the perturbed
    ORACLE runs mark, inside it a multiple of their spread and the north-star marker bound."""
    from moshpp_amd import workload
    from tests import parity_envelope as pe
    job = workload.make_job('smplh', 4000, 53, seed=11)
    solver = workload.make_solver(job)
    seq = solver.solve(job['obs'], job['vis'])
    solved = np.flatnonzero(seq['status'] == 0)
    rep = pe.check(11, seq['pose'][solved], seq['trans'][solved], seq['iters'][solved], frames=solved)
    print(f'seed 11 sequential vs oracle: {rep}')
    assert pe.ok(rep) and rep['ill_conditioned_frames'] > 100, rep
    for _ in range(2):
        chk = solver.solve(job['obs'], job['vis'], chain_mode='chunked', verify_tol=1e-11)
        rs = pe.compare(11, chk, seq)
        print(f'seed 11 chunked vs sequential: {rs}')
        assert rs['frames_outside_tolerance'] == 0, rs
        sq = ((chk['markers_sim'] - seq['markers_sim']) ** 2).sum(-1) * job['vis']
        frame_rmse = np.sqrt(sq.sum(1) / np.maximum(job['vis'].sum(1), 1))
        assert frame_rmse.max() < 5e-3 and np.sqrt((frame_rmse ** 2).mean()) < 1e-3
        assert np.array_equal(chk['status'], seq['status'])


@pytest.mark.gpu
def test_underdetermined_frames_on_the_device(gpu_lib):
    """The singular-normal-matrix case (MANO has no pose prior: BASELINE config 4's model class) on the DEVICE build: one visible
    marker on the first two frames -> 3 data rows for 6 root / translation unknowns.  The kernel's LDL^T meets a non-positive pivot,
    takes the Cauchy step and flags the frame with status -1; the oracle (chumpy's behaviour: dense solve, lstsq on failure) lands
    on another point of the solution set.  Both fit the lone marker exactly and meet again once the frames are determined
    (the emulated twin is tests/test_chain_emulation.py)."""
    from moshpp_amd import capi
    from tests.helpers import oracle_case, device_case
    case = oracle_case('mano', F=6, M=33, seed=9)
    vis = case['vis'].copy()
    vis[:2] = 0
    vis[:2, 5] = 1
    dev = device_case(case, optimize_fingers=True)
    out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], [dict(attach=dev['attach'], obs=case['obs'], vis=vis, first=True)])[0]
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], vis, 'mano', optimize_fingers=True)
    assert out['status'].tolist() == [-1, -1, 0, 0, 0, 0]
    assert out['errs'][:2, 0].max() < 1e-12 and np.asarray(ref['errs']['data'])[:2].max() < 1e-12
    assert np.abs(out['fullpose'][:2] - ref['fullpose'][:2]).max() > 0.1
    assert np.abs(out['fullpose'][2:] - ref['fullpose'][2:]).max() < 1e-2
    assert np.abs(out['fullpose'][4:] - ref['fullpose'][4:]).max() < np.abs(out['fullpose'][2] - ref['fullpose'][2]).max()   # re-converging
    assert np.allclose(out['errs'][2:, 0], np.asarray(ref['errs']['data'])[2:], rtol=1e-2)


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(gpu_lib):
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (torch.distributed.run on 127.0.0.1) and reports
    n_gpus = 2.  MOSHII_BENCH_ONE_GPU=1 puts both ranks on cuda:0 with gloo (one GPU per box here), so the N > 1 code paths of the
    bench -- partition of the fixed jobs, sharded long sequence, max-over-ranks timing -- run; the numbers mean nothing."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MOSHII_BENCH_ONE_GPU='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--frames', '600',
                        '--seeds', '1000,123', '--no-cpu', '--no-stagei', '--no-sequential', '--strong-sequences', '4', '--long-frames', '1500',
                        '--lbs-frames', '200', '--no-config3'], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['value'] > 0
    assert res['strong']['many_sequences']['frames'] == 4 * 600


@pytest.mark.gpu
def test_chunked_solve_is_exact_under_gpu_contention(gpu_lib):
    """The chunk protocol's waits are bounded spins between workgroups that are normally all resident (one per CU).  Here another
    stream keeps the GPU busy with large matrix products while the 4000-frame sequence is solved chunk-parallel, so that chunk
    workgroups start late, out of order, or after their neighbours have given up waiting: whatever the device-side negotiation then
    leaves undone must be caught by the verification rounds -- the stitched result has to equal the sequential chain as it does on
    an idle GPU, with the fused first launch and (MOSHII_NO_FUSE) with host rounds only."""
    import os
    import subprocess
    import sys
    from moshpp_amd import workload
    job = workload.make_job('smplh', 4000, 53, seed=2024)
    solver = workload.make_solver(job)
    seq = solver.solve(job['obs'], job['vis'])
    # the competitor is another PROCESS (torch carries its own HIP runtime: in this process, behind libmoshii's, it finds no device)
    hog = ("import sys, time, torch\n"
           "a = torch.randn(6144, 6144, device='cuda:0')\n"
           "torch.cuda.synchronize(); print('started', flush=True)\n"
           "t0 = time.time(); n = 0\n"
           "while time.time() - t0 < 120:\n"
           "    for _ in range(8): a = torch.tanh(a @ a) * 0.5\n"
           "    torch.cuda.synchronize(); n += 8\n")
    proc = subprocess.Popen([sys.executable, '-c', hog], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        line = proc.stdout.readline()
        assert line.strip() == 'started', (line, proc.stderr.read() if proc.poll() is not None else '')
        for env in (None, 'MOSHII_NO_FUSE'):
            if env:
                os.environ[env] = '1'
            try:
                for _ in range(2):
                    chk = solver.solve(job['obs'], job['vis'], chain_mode='chunked', verify_tol=1e-9)
                    rep = chk['chunk_report']
                    dp = np.abs(chk['fullpose'] - seq['fullpose']).max()
                    print(f'contention ({env or "fused"}): repaired {rep["n_repaired"]} in {rep["repair_rounds"]} rounds, max|chunked - sequential| {dp:.2e} rad')
                    assert rep['max_handoff_dev'] <= rep['verify_tol']
                    assert dp < 5e-9 and np.array_equal(chk['status'], seq['status'])
            finally:
                if env:
                    del os.environ[env]
        assert proc.poll() is None          # the competing process really ran alongside the whole time
    finally:
        proc.kill()
        proc.wait()


@pytest.mark.gpu
def test_chunks_given_up_in_the_tail_of_the_first_launch_are_resolved_exactly(gpu_lib):
    """Round 6: the first launch of a chunked solve used to last as long as its slowest chunk (250 chunks of the bench sequence: the median
    needs 13.4 ms, the launch took 21.8).  Now a chain that still has frames to go when all but a fifth of a chip's worth of the others
    have ended gives its chunk up (ChainDev::tail_done): spoiled hand-off states, a mark that no sweep re-joins inside the chunk, and the
    host's rounds re-solve it from its predecessor's end state.  Here with the quota forced low (MOSHII_TAIL_CUT=62: chains give up as soon
    as 188 of 250 are done -- a fifth of the chunks), in a process of its own (the variable is read once): the repair trace must show
    given-up chunks (deviation code 5e+299), and the stitched result must be the sequential chain's as always."""
    import os
    import subprocess
    import sys
    code = ("import sys\nsys.path.insert(0, '.')\nimport numpy as np\nfrom moshpp_amd import workload\n"
            "worst = 0.0\n"
            "for seed in (1000, 5):\n"
            "    job = workload.make_job('smplh', 4000, 53, seed=seed); solver = workload.make_solver(job)\n"
            "    seq = solver.solve(job['obs'], job['vis'], chain_mode='sequential')\n"
            "    for _ in range(2):\n"
            "        chk = solver.solve(job['obs'], job['vis'], chain_mode='chunked', verify_tol=1e-9)\n"
            "        assert np.array_equal(chk['status'], seq['status']) and np.array_equal(chk['iters'][:, 0] > 0, seq['iters'][:, 0] > 0)\n"
            "        worst = max(worst, float(np.abs(chk['fullpose'] - seq['fullpose']).max()), float(np.abs(chk['trans'] - seq['trans']).max()))\n"
            "        print('REPORT', seed, chk['chunk_report']['n_repaired'], chk['chunk_report']['repair_rounds'], chk['chunk_report']['max_handoff_dev'], flush=True)\n"
            "print('WORST', worst, flush=True)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MOSHII_TAIL_CUT='62', MOSHII_TRACE_REPAIR='1')
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    worst = float([l for l in p.stdout.splitlines() if l.startswith('WORST')][0].split()[1])
    given_up = p.stderr.count(':5e+299:')
    print(p.stdout.strip().replace('\n', ' | '), f'| chunks given up (trace): {given_up}')
    assert given_up >= 8, p.stderr[-2000:]          # the path was taken, many times
    assert worst < 5e-9, worst


@pytest.mark.gpu
def test_default_cooperative_chain_beside_a_competing_process(gpu_lib):
    """The drop-in default -- ONE sequential chain on six workgroups that wait for each other through device memory -- while another
    process keeps the GPU busy (how the reference is deployed: one process per capture, several per device, mosh_head.py:584-589):
    (a) a matrix-product hog on another stream of another process, (b) a second MoSh process solving its own sequence at the same
    time.  The result must be the one-workgroup chain's to round-off with the same iteration counts -- whether the group stayed whole
    or gave up and the library repeated the call with plain chains (then it says so on stderr and stays with plain chains) -- and the
    wall time must stay within 3x of the idle one plus the one-off price of a broken group (~0.1 s wait + the repeated solve)."""
    import os
    import subprocess
    import sys
    import time
    from moshpp_amd import workload
    F = 1500
    job = workload.make_job('smplh', F, 53, seed=2024)
    solver = workload.make_solver(job)
    plain = solver.solve(job['obs'], job['vis'], coop_group=1)   # one workgroup per chain (the bit-reproducible reference of this test)
    t0 = time.perf_counter(); idle = solver.solve(job['obs'], job['vis']); t_idle = time.perf_counter() - t0
    assert np.abs(idle['fullpose'] - plain['fullpose']).max() < 1e-8 and np.array_equal(idle['iters'], plain['iters'])
    hog = ("import sys, time, torch\n"
           "a = torch.randn(6144, 6144, device='cuda:0')\n"
           "torch.cuda.synchronize(); print('started', flush=True)\n"
           "t0 = time.time()\n"
           "while time.time() - t0 < 60:\n"
           "    for _ in range(8): a = torch.tanh(a @ a) * 0.5\n"
           "    torch.cuda.synchronize()\n")
    other = ("import sys, time\nsys.path.insert(0, '.')\nimport numpy as np\nfrom moshpp_amd import workload\n"
             "job = workload.make_job('smplh', 1500, 53, seed=7); solver = workload.make_solver(job)\n"
             "solver.solve(job['obs'][:50], job['vis'][:50]); print('started', flush=True)\n"
             "t0 = time.time(); n = 0\n"
             "while time.time() - t0 < 25:\n"
             "    o = solver.solve(job['obs'], job['vis']); n += 1\n"
             "print('solved', n, float(np.abs(o['fullpose']).max()), flush=True)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, code in (('matrix-product hog', hog), ('second MoSh process', other)):
        proc = subprocess.Popen([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root)
        try:
            line = proc.stdout.readline()
            assert line.strip() == 'started', (line, proc.stderr.read() if proc.poll() is not None else '')
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); out = solver.solve(job['obs'], job['vis']); ts.append(time.perf_counter() - t0)
                assert np.abs(out['fullpose'] - plain['fullpose']).max() < 1e-8 and np.abs(out['trans'] - plain['trans']).max() < 1e-8
                assert np.array_equal(out['iters'], plain['iters']) and np.array_equal(out['status'], plain['status'])
            print(f'cooperative default beside a {name}: {[round(t, 3) for t in ts]} s against {t_idle:.3f} s idle')
            assert proc.poll() is None, 'the competing process ended before the solves did'
            # a broken group costs its wait limit and the repeated solve ONCE (the fallback is sticky); the other runs are plain solves
            assert sorted(ts)[1] < 3.0 * t_idle + 0.1 and max(ts) < 3.0 * t_idle + 0.6, (ts, t_idle)
        finally:
            proc.kill()
            proc.wait()


@pytest.mark.gpu
def test_config3_oracle_window_on_three_captures(gpu_lib):
    """BASELINE configs[2]'s subject as bench.py's `config3` leg builds it (workload.make_face_job: SMPL-X, 89 markers incl. face / hand
    vertices, fingers + jaw + 80 expression coefficients free: 194 unknowns), the captures 7000, 7001, 7002 (the first three motion
    seeds of the bench leg: nobody picked them), 400 frames each, against the committed oracle trajectories
    (tests/golden/make_config3_golden.py): equal dogleg iteration counts on every frame, states to 1e-6 on every 10th.  Both as the
    library runs the chain by default (cooperative) and as one workgroup."""
    from moshpp_amd import capi, workload
    from tests.helpers import face_capture_host, face_job_oracle
    job = workload.make_face_job()
    solver = workload.make_solver(job)
    m, pr, closest, coef = face_job_oracle(job)
    assert np.array_equal(closest, solver.tc.closest) and np.abs(coef - solver.tc.coef).max() < 1e-12
    g = _config3_golden()
    H = int(g['frames'])
    for ms in (7000, 7001, 7002):
        cap = face_capture_host(job, m, closest, coef, ms, H)
        ch = [dict(attach=solver.attach, obs=cap['obs'], vis=cap['vis'], first=True)]
        for coop in (0, 1):
            out = capi.chain_solve_host(solver.dev, solver.prior, solver.opts, ch, coop=coop)[0]
            assert np.all(out['status'] == 0)
            dp, ds, dtr, same = _hold_to_config3_golden(g, ms, out, H)
            d = (out['markers_sim'] - cap['obs'])[cap['vis']]
            print(f'config 3 capture {ms} ({capi.last_launch_info()[0]}): {H} frames vs oracle {dp:.2e} rad / {ds:.2e} (expression); iteration counts equal on '
                  f'{same.mean() * 100:.1f} % of frames; marker rmse {np.sqrt((d ** 2).sum(1).mean()):.2e} m; data SSE max {out["errs"][:, 0].max():.1f}')
            assert dp < 1e-6 and ds < 1e-6 and same.all()
