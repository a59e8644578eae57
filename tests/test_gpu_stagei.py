"""Stage-I on the GPU (moshii_stagei_solve) against the f64 oracle (oracle/stagei_oracle.py) on the same seeded problems."""
import os
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
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stagei_golden.npz'))      # committed oracle output
    gname = 'fingers' if fingers else 'body'
    assert np.abs(out['betas'] - G[f'{gname}_betas']).max() < 1e-5 and np.abs(out['markers_latent'] - G[f'{gname}_markers_latent']).max() < 1e-6
    e = ref['errs']
    want = dict(data=e['data'], poseB=e['poseB'], init=e['init_0'], beta=e['beta'], surf=e['surf'], poseH=e.get('poseH', 0.0))
    for k, v in want.items():
        assert abs(out['errs'][k] - v) <= 1e-5 * max(1.0, abs(v)), k


def _variant(name):
    """(case, optimize_fingers, extra arguments) of the Stage-I parity variants."""
    if name == 'smplx_exclude':      # SMPL-X topology; a vertex set the attachment may not use (the eyeballs of transformed_lm.py:49-50)
        c = helpers.stagei_case('smplx', seed=3)
        dom = np.argmax(c['dd']['weights'], 1)
        return c, False, dict(exclude_vids=np.flatnonzero(dom >= 23)[:200])
    if name == 'smpl':
        return helpers.stagei_case('smpl', seed=4), False, {}
    if name == 'mano':               # no body prior, only root + fingers
        return helpers.stagei_case('mano', n_verts=700, M=14, seed=5, nb=4), True, {}
    if name == 'fixed_betas':        # optimize_betas = false: no shape unknowns, no beta term
        return helpers.stagei_case('smplh', seed=6, nb=0), False, {}
    if name == 'betas_init':         # betas_fname given and optimize_betas = true: the solve starts from them (chmosh.py:164-170)
        return helpers.stagei_case('smplh', seed=7), False, dict(betas_init=np.array([0.3, -0.2, 0.1, 0.0, 0.4, -0.5]))
    if name == 'head_corr':          # head-marker correlation term (chmosh.py:252-266, 362-369)
        rng = np.random.default_rng(0)
        return helpers.stagei_case(), False, dict(head_corr=(np.array([3, 7, 11, 20]), rng.normal(0, 1, (3, 4))))
    if name == 'face':               # SMPL-X, fixed betas, jaw + per-frame expression coefficients free in the last two rounds
        return helpers.stagei_case('smplx', seed=3, nb=0), False, dict(n_expr=5, expr_start=4, face_ids=[66, 67, 68])
    if name == 'extra_rigid':        # opt_settings.extra_initial_rigid_adjustment (chmosh.py:230-232)
        return helpers.stagei_case('smplh', seed=8), False, dict(extra_initial_rigid_adjustment=True)
    if name == 'collinear':
        # The collinear-neighbour fallback of the live attachment (transformed_lm.py:94-101): one marker's three nearest canonical
        # vertices are made EXACTLY collinear -- the layout vertex and two vertices moved next to it along x, all three with the same
        # y / z template coordinates, blend-shape rows, pose-corrective rows and skinning weights, so that their canonical y / z stay
        # bit-identical for every betas in every implementation (at zero pose the skinning transforms are exact identities) and
        # e1 x e2 is an exact zero.  The reference then moves the third neighbour of EVERY marker to the fourth nearest.
        import copy
        from oracle import stageii_oracle as so_
        c = copy.deepcopy(helpers.stagei_case('smplh', seed=9))
        mdl = c['model']
        v = mdl['v_template']
        v0 = int(c['vids'][0])
        d = np.linalg.norm(v - v[v0], axis=1)
        near = np.argsort(d)[1:3]
        delta = 0.25 * d[near[0]]
        for k, vi in enumerate(near):
            v[vi] = v[v0] + np.array([(1.0, -1.3)[k] * delta, 0.0, 0.0])
            for key in ('shapedirs', 'posedirs', 'weights'):
                mdl[key][vi] = mdl[key][v0]
        c['m'] = so_.prepare_model(mdl)
        so_.set_free_shape(c['m'], 0, c['nb'])
        can = so_.verts_forward(c['m'], so_.fullpose_from_pose(c['m'], np.zeros(c['m']['NP'])), np.zeros(3))
        e1, e2 = can[near[0]] - can[v0], can[near[1]] - can[v0]
        assert np.all(np.cross(e1, e2) == 0.0)
        return c, False, {}
    raise KeyError(name)


@pytest.mark.parametrize('name', ['smplx_exclude', 'smpl', 'mano', 'fixed_betas', 'betas_init', 'head_corr', 'face', 'extra_rigid', 'collinear'])
def test_stagei_variants_match_oracle(name):
    from moshpp_amd import capi
    from oracle import stagei_oracle as s1
    c, fingers, extra = _variant(name)
    mdl = c['model']
    dev = capi.Model(mdl['v_template'], mdl['shapedirs'], mdl['posedirs'], mdl['weights'], mdl['J_regressor'], mdl['parents'],
                     mdl['body_dof'], mdl['hand_dof'], mdl['hands_mean'], mdl['selected_components'])
    pr = capi.Prior(c['prior']['means'], c['prior']['chols'], c['prior']['weights']) if c['prior'] is not None else None
    out = capi.stagei_solve_host(dev, pr, **helpers.stagei_kwargs(c, optimize_fingers=fingers, **extra))
    ref = s1.stagei_solve(c['m'], c['faces'], c['prior'], c['model_type'], c['frames'], c['vids'], c['mask'], c['m2b'], c['nb'],
                          optimize_fingers=fingers, **helpers.stagei_oracle_extra(extra))
    if name == 'face':
        assert np.abs(out['expression'] - ref['expression']).max() < 1e-5 and np.abs(ref['expression']).max() > 1e-3
    if c['nb']:
        assert np.abs(out['betas'] - ref['betas']).max() < 1e-5
    assert np.abs(out['markers_latent'] - ref['markers_latent']).max() < 1e-6
    assert np.abs(out['pose'] - ref['pose']).max() < 1e-5 and np.abs(out['trans'] - ref['trans']).max() < 1e-6
    assert (out['markers_latent_vids'] == ref['markers_latent_vids']).all()
    if name == 'head_corr':
        assert abs(out['errs']['init_head_corr'] - ref['errs']['init_head_corr']) < 1e-6 * max(1.0, ref['errs']['init_head_corr'])


def test_stagei_rejects_bad_input():
    from moshpp_amd import capi
    c = helpers.stagei_case()
    dev, pr = _device(c)
    kw = helpers.stagei_kwargs(c)
    kw['frames'] = [(np.array([0, 1, c['M'] + 3]), np.zeros((3, 3)))] + list(kw['frames'][1:])
    with pytest.raises(capi.MoshiiError):
        capi.stagei_solve_host(dev, pr, **kw)


def test_mosh_stagei_then_stageii_end_to_end(tmp_path):
    """Files in, files out: model pickle with faces, priors, marker-layout json, two npz captures -> frame picker -> mosh_stagei
    (GPU) -> mosh_stageii (GPU) through the reference's plugin signatures; Stage-I checked against the oracle on the same picked frames."""
    import json
    import os
    import pickle
    from moshpp_amd import synth
    from moshpp_amd.cfg import make_cfg
    from moshpp_amd.mosh_head import run_stagei, run_stageii
    from oracle import stageii_oracle as so
    from oracle import stagei_oracle as s1
    c = helpers.stagei_case(M=32, F=4, seed=5)
    dd, m, M, nb = c['dd'], c['m'], c['M'], c['nb']
    raw = {k: v for k, v in dd.items() if not k.startswith('_')}
    with open(tmp_path / 'model.pkl', 'wb') as f:
        pickle.dump(raw, f)
    with open(tmp_path / 'pose_body_prior.pkl', 'wb') as f:
        pickle.dump(synth.synth_gmm_prior(5), f)
    np.savez(tmp_path / 'pose_hand_prior.npz', **synth.synth_hand_prior(5))
    labels = [f'MK{i:02d}' for i in range(M)]
    layout = {'surface_model_type': 'smplh', 'markersets': [
        {'type': 'body', 'distance_from_skin': 0.0095, 'indices': {l: int(v) for l, v in zip(labels[:M - 4], c['vids'][:M - 4])}},
        {'type': 'head', 'distance_from_skin': 0.0095, 'indices': {l: int(v) for l, v in zip(labels[M - 4:], c['vids'][M - 4:])}}]}
    with open(tmp_path / 'layout.json', 'w') as f:
        json.dump(layout, f)
    # a capture: ground-truth motion of the case's subject, 60 frames, a few dropouts
    cl, coef = so.transformed_coeffs(so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3), None, shp=c['betas_gt']),
                                     c['ml_gt'])
    pose_gt, trans_gt = synth.synth_motion(m['NP'], m['body_dof'], 60, seed=5)
    rng = np.random.default_rng(8)
    mk = np.zeros((60, M, 3))
    for t in range(60):
        p = pose_gt[t].copy(); p[m['body_dof']:] = 0; p[30:36] = 0
        vv = so.verts_forward(m, so.fullpose_from_pose(m, p), trans_gt[t], cl.reshape(-1), shp=c['betas_gt']).reshape(M, 3, 3)
        mk[t] = so.markers_from_verts(coef, vv[:, 0], vv[:, 1], vv[:, 2]) + rng.normal(0, 0.0003, (M, 3))
    mk[rng.random((60, M)) < 0.03] = np.nan
    cap = str(tmp_path / 'capture.npz')
    np.savez(cap, markers=mk * 1000.0, labels=np.array(labels), frame_rate=120.0)
    cfg = make_cfg(**{'mocap.fname': cap, 'surface_model.type': 'smplh', 'surface_model.fname': str(tmp_path / 'model.pkl'),
                      'surface_model.num_betas': nb, 'surface_model.dof_per_hand': 12, 'surface_model.use_hands_mean': False,
                      'moshpp.pose_body_prior_fname': str(tmp_path / 'pose_body_prior.pkl'),
                      'moshpp.pose_hand_prior_fname': str(tmp_path / 'pose_hand_prior.npz'),
                      'dirs.marker_layout.fname': str(tmp_path / 'layout.json'),
                      'moshpp.stagei_frame_picker.num_frames': 5, 'moshpp.stagei_frame_picker.least_avail_markers': 0.9})
    stagei = run_stagei(cfg, [cap], stagei_fname=str(tmp_path / 'out' / 'stagei.pkl'))
    assert set(stagei) >= {'betas', 'markers_latent', 'latent_labels', 'marker_meta', 'markers_latent_vids', 'stagei_debug_details'}
    assert stagei['latent_labels'] == labels and stagei['betas'].shape == (10,) and stagei['markers_latent'].shape == (M, 3)
    dbg = stagei['stagei_debug_details']
    assert list(dbg['stagei_errs']) == ['data', 'poseB', 'init_body', 'init_head', 'beta', 'surf'] and len(dbg['stagei_fnames']) == 5
    assert os.path.exists(tmp_path / 'out' / 'stagei.pkl')
    assert len(dbg['stagei_markers_sim_all']) == 5 and dbg['stagei_markers_sim_all'][0].shape == (M, 3)
    assert all(a.shape == b.shape for a, b in zip(dbg['stagei_markers_sim'], dbg['stagei_markers_obs']))
    fit = np.sqrt(np.mean([((a - b) ** 2).sum(1).mean() for a, b in zip(dbg['stagei_markers_sim'], dbg['stagei_markers_obs'])]))
    assert fit < 5e-3 and set(dbg['markers_latent_all_vids']) <= set(labels)
    # oracle on the same picked frames
    frames = []
    for fr in dbg['stagei_frames']:
        common = [l for l in labels if l in fr and not np.any(np.isnan(fr[l]))]
        frames.append((np.array([labels.index(l) for l in common]), np.vstack([fr[l] for l in common])))
    mm = stagei['marker_meta']
    ref = s1.stagei_solve(m, dd['f'], c['prior'], 'smplh', frames, np.array(list(mm['marker_vids'].values())), mm['marker_type_mask'],
                          mm['m2b_distance'], nb)
    assert np.abs(stagei['betas'][:nb] - ref['betas']).max() < 1e-5
    assert np.abs(stagei['markers_latent'] - ref['markers_latent']).max() < 1e-6
    # Stage-II with the Stage-I result, through the head function (merges the Stage-I keys)
    cfg.mocap.end_fidx = 20
    stageii = run_stageii(stagei, cfg, stageii_fname=str(tmp_path / 'out' / 'capture_stageii.pkl'))
    assert stageii['fullpose'].shape == (20, 156) and 'markers_latent' in stageii
    rm = np.sqrt(np.mean([((a - b) ** 2).sum(1).mean() for a, b in zip(stageii['stageii_debug_details']['markers_sim'],
                                                                     stageii['stageii_debug_details']['markers_obs'])]))
    assert rm < 5e-3        # the solved subject + layout reproduce the capture to a few millimetres


@pytest.mark.parametrize('seed', [1, 2, 3, 4, 5])
def test_stagei_schur_solver_matches_dense(monkeypatch, seed):
    """The default arrow-structured solver (per-frame elimination + Schur complement of the shared block) takes the same Gauss-Newton
    steps as the dense blocked Cholesky (MOSHII_S1_SOLVER=dense), on five seeded problems (the bench-sized ones:
    profiles/r02_stagei_schur_seeds.txt)."""
    from moshpp_amd import capi
    c = helpers.stagei_case(seed=seed)
    dev, pr = _device(c)
    kw = helpers.stagei_kwargs(c)
    monkeypatch.setenv('MOSHII_S1_SOLVER', 'dense')
    a = capi.stagei_solve_host(dev, pr, **kw)
    monkeypatch.delenv('MOSHII_S1_SOLVER', raising=False)
    b = capi.stagei_solve_host(dev, pr, **kw)
    assert a['iters'] == b['iters']
    assert np.abs(a['betas'] - b['betas']).max() < 1e-6 and np.abs(a['markers_latent'] - b['markers_latent']).max() < 1e-7
    assert np.abs(a['pose'] - b['pose']).max() < 1e-6


def _nccl_stagei_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world)
    from moshpp_amd import capi
    from moshpp_amd.parallel import stagei_solve_sharded
    c = helpers.stagei_case(M=24, F=5, n_verts=1500, seed=11)
    dev, pr = _device(c)
    kw = helpers.stagei_kwargs(c)
    single = capi.stagei_solve_host(dev, pr, **kw)
    out = stagei_solve_sharded(lambda **sh: capi.stagei_solve_host(dev, pr, **kw, **sh), len(c['frames']), dist)   # nccl -> on device
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), iters=[out['iters'], single['iters']], betas=out['betas'], betas1=single['betas'],
             ml=out['markers_latent'], ml1=single['markers_latent'], pose=out['pose'], pose1=single['pose'])
    dist.barrier()
    dist.destroy_process_group()


def test_stagei_allreduce_on_device_through_rccl(tmp_path):
    """moshii_stagei_desc.allreduce_on_device with torch.distributed's nccl backend (= RCCL): the callback wraps the solver's device
    buffers as tensors and all-reduces them in place.  One rank (this box has one GPU): the collective is the identity, what is
    tested is that RCCL accepts the foreign pointers, the stream hand-over and that the sharded code path returns the plain result;
    the multi-rank arithmetic is covered with gloo in tests/test_distributed_gloo.py."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_nccl_stagei_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    o = np.load(tmp_path / 'rank0.npz')
    assert int(o['iters'][0]) == int(o['iters'][1])
    assert np.abs(o['betas'] - o['betas1']).max() < 1e-9 and np.abs(o['ml'] - o['ml1']).max() < 1e-10
    assert np.abs(o['pose'] - o['pose1']).max() < 1e-9


@pytest.mark.parametrize('name', ['smplh_body', 'smplh_extra_rigid', 'smplh_fingers', 'smplh_head_corr', 'smplh_fixed_betas',
                                  'smplh_betas_init', 'smpl_body', 'mano_fingers', 'smplx_face'])
def test_mosh_stagei_matches_executed_reference(name, tmp_path):
    """The drop-in `mosh_stagei` (files in, dict out; kernels on the GPU) against the reference's own `mosh_stagei` EXECUTED on the
    same files and frame dicts (tests/golden/make_ref_stagei_golden.py -> ref_stagei.npz): result keys, the per-frame label matching,
    betas / latent markers / poses / translations, nearest vertex ids, and stagei_errs under the reference's keys in its order."""
    from moshpp_amd import chmosh
    from moshpp_amd.cfg import make_cfg
    from tests.test_ref_golden import stagei_ref_case, check_stagei_against_reference_run
    sc = stagei_ref_case(name, tmp_path)
    c, ref = sc['case'], sc['ref']
    cfg = make_cfg(**{'surface_model.type': sc['model_type'], 'surface_model.fname': c['model_fname'], 'surface_model.num_betas': sc['nb_cfg'],
                      'surface_model.dof_per_hand': c['dof_per_hand'], 'surface_model.use_hands_mean': False,
                      'moshpp.pose_body_prior_fname': c['body_prior_fname'], 'moshpp.pose_hand_prior_fname': c['hand_prior_fname'],
                      'moshpp.optimize_fingers': sc['fingers'], 'moshpp.optimize_betas': sc['optimize_betas'],
                      'moshpp.head_marker_corr_fname': c['head_corr_fname'], 'moshpp.optimize_face': sc['face'],
                      'surface_model.num_expressions': sc['n_expr'], 'surface_model.betas_expr_start_id': sc['expr_start'],
                      'opt_settings.extra_initial_rigid_adjustment': sc['extra'], 'dirs.marker_layout.fname': c['layout_fname'],
                      'opt_settings.weights_type': 'smplh'})   # (the yaml has no smpl / mano tables: the fixture used the smplh weights)
    res = chmosh.mosh_stagei(c['frames'], cfg, betas_fname=c['betas_fname'])
    assert sorted(res) == list(ref[f'{name}_keys'])
    assert bool(cfg.moshpp.optimize_fingers) == bool(ref[f'{name}_optimize_fingers_after'])
    assert res['latent_labels'] == list(ref[f'{name}_latent_labels'])
    dbg = res['stagei_debug_details']
    assert set(ref[f'{name}_debug_keys']) <= set(dbg)
    assert ['|'.join(sorted(l)) for l in dbg['stagei_labels_obs']] == list(ref[f'{name}_labels_obs'])
    got = dict(betas=res['betas'][:sc['nb']], markers_latent=res['markers_latent'],
               markers_latent_vids=[res['markers_latent_vids'][l] for l in res['latent_labels']],
               pose=np.array(dbg['opt_models_pose']), trans=np.array(dbg['opt_models_trans']), errs=dbg['stagei_errs'])
    assert np.all(res['betas'][sc['nb']:] == 0)
    check_stagei_against_reference_run(name, ref, got)
