"""Shared builders for the tests: the same seeded synthetic case prepared for the oracle (numpy f64)
and for libmoshii (device handles)."""
import numpy as np

from moshpp_amd import synth
from oracle import stageii_oracle as so


def pose_layout(s):
    """(body_dof, hand_dof, hands_mean, selected_components) -- smpl_fast_derivatives.py:80-128."""
    mt = s['model_type']
    dd = s['model']
    K = synth.MODEL_DIMS[mt][1]
    if mt in ('smplh', 'smplx'):
        d = s['dof_per_hand']
        hp = s['hand_prior']
        comps = np.zeros((2 * d, 90))
        comps[:d, :45] = hp['componentsl'][:d]
        comps[d:, 45:] = hp['componentsr'][:d]
        hm = np.concatenate([hp['hands_meanl'], hp['hands_meanr']]) if s['use_hands_mean'] else np.zeros(90)
        return 3 * K - 90, 2 * d, hm, comps
    if mt == 'mano':
        d = s['dof_per_hand']
        hm = np.zeros(45) if s['use_hands_mean'] else dd['hands_mean']
        return 3, d, hm, dd['hands_components'][:d]
    return 3 * K, 0, None, None


def oracle_case(model_type='smplh', F=12, M=53, seed=0, **kw):
    s = synth.make_sequence(model_type, F, M, seed=seed, **kw)
    dd = s['model']
    bd, hd, hm, comps = pose_layout(s)
    model = dict(v_template=dd['v_template'], shapedirs=dd['shapedirs'], posedirs=dd['posedirs'],
                 weights=dd['weights'], J_regressor=dd['J_regressor'], parents=synth.kintree_parents(model_type),
                 body_dof=bd, hand_dof=hd, hands_mean=hm, selected_components=comps)
    m = so.prepare_model(model, s['betas'])
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3))
    closest, coef = so.transformed_coeffs(can, s['markers_latent'])
    npose = 63 if model_type in ('smplh', 'smplx') else 69
    prior = so.prepare_gmm_prior(s['gmm'], npose) if model_type != 'mano' else None
    obs = np.nan_to_num(s['markers'])
    vis = ~np.isnan(s['markers']).any(-1)
    return dict(s=s, m=m, model=model, can=can, closest=closest, coef=coef, prior=prior, obs=obs, vis=vis,
                model_type=model_type)


def device_case(case, optimize_fingers=False, optimize_toes=False, maxiter=100, weights=None, optimize_face=False,
                shape_kind=None):
    """libmoshii handles + options for an oracle_case / shape_case (same arrays, same ids)."""
    from moshpp_amd import capi
    mdl = case['model']
    m = case['m']
    dev = capi.Model(mdl['v_template'], mdl['shapedirs'], mdl['posedirs'], mdl['weights'], mdl['J_regressor'],
                     mdl['parents'], mdl['body_dof'], mdl['hand_dof'], mdl['hands_mean'], mdl['selected_components'])
    dev.set_betas(case['s']['betas'])
    if shape_kind is not None:
        dev.set_free_shape(case['start'], case['E'])          # before the attachment: it gathers its rows of the block
    att = capi.Attachment(dev, case['closest'], case['coef'])
    pr = None
    if case['prior'] is not None:
        pr = capi.Prior(case['prior']['means'], case['prior']['chols'], case['prior']['weights'])
    root, body, finger, st1, st2 = so.pose_id_sets(case['model_type'], m['NP'], optimize_fingers, optimize_toes,
                                                   optimize_face=optimize_face)
    W = so.stageii_weights_default() if weights is None else weights
    opts = capi.make_opts(W, st1, st2, body, finger if optimize_fingers else [], maxiter=maxiter,
                          face_ids=so.face_pose_ids(case['model_type'], optimize_face),
                          n_shape=case['E'] if shape_kind is not None else 0, shape_kind=shape_kind)
    return dict(model=dev, attach=att, prior=pr, opts=opts)


def shape_case(model_type='smplx', F=6, M=40, E=6, seed=0, kind='expr', boost=6.0):
    """A case whose observations carry time-varying FREE shape coefficients (expression: `kind='expr'`, with jaw motion;
    DMPL: `kind='dmpl'`): shapedirs gets E extra columns [16, 16+E) that Stage-II may move per frame
    (chmosh.py:507-514, 562-567, 685-699).  Returns an oracle_case-like dict (+ shp_gt, E, start)."""
    NB = 16 + E
    dd = dict(synth.synth_model(model_type, seed=seed, num_betas=NB))
    sd = np.array(dd['shapedirs'], dtype=np.float64)
    sd[:, :, 16:] *= boost / np.maximum(np.abs(sd[:, :, 16:]).max(axis=(0, 1), keepdims=True) / 0.005, 1e-12)
    dd['shapedirs'] = sd
    s = synth.make_sequence(model_type, F, M, seed=seed, num_betas=NB, dd=dd, body_only_markers=False, dropout=0.0,
                            n_gaps=0)
    s['betas'] = s['betas'].copy()
    s['betas'][16:] = 0.0
    bd, hd, hm, comps = pose_layout(s)
    model = dict(v_template=dd['v_template'], shapedirs=sd, posedirs=dd['posedirs'], weights=dd['weights'],
                 J_regressor=dd['J_regressor'], parents=synth.kintree_parents(model_type), body_dof=bd, hand_dof=hd,
                 hands_mean=hm, selected_components=comps)
    m = so.prepare_model(model, s['betas'])
    so.set_free_shape(m, 16, E)
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3))
    closest, coef = so.transformed_coeffs(can, s['markers_latent'],
                                          exclude_vids=np.arange(9383, 10475) if model_type == 'smplx' else None)
    rng = np.random.default_rng(seed + 77)
    tt = np.arange(F)[:, None] / 30.0
    shp_gt = 1.2 * np.sin(2 * np.pi * (0.7 + 0.3 * rng.random(E))[None] * tt + rng.random(E)[None] * 6.0)
    pose_gt = s['pose_gt'].copy()
    pose_gt[:, bd:] = 0.0
    if kind == 'expr' and model_type == 'smplx':
        pose_gt[:, 66:69] = 0.15 * np.sin(2 * np.pi * 1.1 * tt + np.array([0.0, 1.0, 2.0]))   # jaw
    obs = np.zeros((F, M, 3))
    flat = closest.reshape(-1)
    for f in range(F):
        v = so.verts_forward(m, so.fullpose_from_pose(m, pose_gt[f]), s['trans_gt'][f], flat, shp=shp_gt[f])
        v = v.reshape(M, 3, 3)
        obs[f] = so.markers_from_verts(coef, v[:, 0], v[:, 1], v[:, 2])
    obs += rng.normal(0, 0.0003, obs.shape)
    vis = np.ones((F, M), dtype=bool)
    vis[1:, 3] = False                      # one missing marker: annealing factor != 1
    npose = 63 if model_type in ('smplh', 'smplx') else 69
    prior = so.prepare_gmm_prior(s['gmm'], npose)
    return dict(s=s, m=m, model=model, can=can, closest=closest, coef=coef, prior=prior, obs=obs, vis=vis,
                model_type=model_type, shp_gt=shp_gt, pose_gt=pose_gt, E=E, start=16, kind=kind)


# ---- Stage-I ----------------------------------------------------------------------------------------------------
def stagei_case(model_type='smplh', n_verts=2500, nb=6, M=30, F=6, seed=0, dof_per_hand=12, finger_markers=False):
    """synth.make_stagei_problem prepared for the oracle: `m` (prepare_model + free shape block) and the prepared GMM prior."""
    pb = synth.make_stagei_problem(model_type, n_verts=n_verts, nb=nb, M=M, F=F, seed=seed, dof_per_hand=dof_per_hand,
                                   finger_markers=finger_markers)
    m = so.prepare_model(pb['model'])
    so.set_free_shape(m, 0, nb)
    npose = 63 if model_type in ('smplh', 'smplx') else 69
    prior = so.prepare_gmm_prior(pb['gmm'], npose) if model_type != 'mano' else None
    return dict(m=m, model=pb['model'], faces=pb['faces'], prior=prior, frames=pb['frames'], vids=pb['vids'], betas_gt=pb['betas_gt'],
                ml_gt=pb['ml_gt'], nb=nb, M=M, mask={'body': np.ones(M, bool)}, m2b={'body': pb['skin']}, model_type=model_type,
                dd=pb['dd'], problem=pb)


def stagei_oracle_extra(extra):
    """The oracle's spelling of the optional Stage-I arguments (n_expr / expr_start / face_ids -> optimize_face ...)."""
    e = {k: v for k, v in extra.items() if k in ('exclude_vids', 'head_corr', 'betas_init', 'extra_initial_rigid_adjustment')}
    if extra.get('n_expr'):
        e.update(optimize_face=True, expr_start=extra['expr_start'], n_expr=extra['n_expr'])
    return e


def stagei_kwargs(case, optimize_fingers=False, exclude_vids=None, head_corr=None, betas_init=None, n_expr=0, expr_start=0, face_ids=(),
                  extra_initial_rigid_adjustment=False):
    """The arguments of capi.stagei_desc for a stagei_case (reference default weights)."""
    from oracle import stagei_oracle as s1
    m = case['m']
    root, body, finger, step1, _ = so.pose_id_sets(case['model_type'], m['NP'], optimize_fingers=optimize_fingers)
    if case['model_type'] == 'mano' and not optimize_fingers:
        finger = []                       # chmosh.py:300-301 sets the ids, :390-393 adds them only with optimize_fingers
    M = case['M']
    W = s1.stagei_weights_default()
    return dict(faces=case['faces'], marker_vids=case['vids'], m2b=np.ones(M) * case['m2b']['body'], wt_init=np.ones(M) * W['stagei_wt_init'],
                frames=case['frames'], nb=case['nb'], weights=W, pose_ids=step1, body_ids=body if case['prior'] is not None else [],
                finger_ids=finger, exclude_vids=exclude_vids, head_corr=head_corr, betas_init=betas_init, n_expr=n_expr,
                expr_start=expr_start, face_ids=face_ids, extra_initial_rigid_adjustment=extra_initial_rigid_adjustment)


# ---- BASELINE config 3 (SMPL-X, face + fingers free) ---------------------------------------------------------------------------------
def face_job_oracle(job):
    """The oracle's model / prior / attachment of a workload.make_face_job subject."""
    sm = job['sm']
    E = job['num_expressions']
    m = so.prepare_model(dict(v_template=sm.v_template, shapedirs=sm.shapedirs, posedirs=sm.posedirs, weights=sm.weights,
                              J_regressor=sm.J_regressor, parents=sm.parents, body_dof=sm.body_dof, hand_dof=sm.hand_dof,
                              hands_mean=sm.hands_mean, selected_components=sm.selected_components), job['betas'])
    so.set_free_shape(m, job['betas_expr_start_id'], E)
    pr = so.prepare_gmm_prior(job['seq']['gmm'], 63)
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3))
    closest, coef = so.transformed_coeffs(can, job['markers_latent'], exclude_vids=np.arange(9383, 10475))
    return m, pr, closest, coef


def face_capture_host(job, m, closest, coef, motion_seed, n_frames, expr_amp=0.3, noise=0.0005, dropout=0.02):
    """A capture of the config-3 subject generated on the HOST (oracle forward; the same recipe as workload.make_face_capture, which
    goes through the device): seeded body + finger motion, a jaw motion, a per-capture expression offset, noise and dropout.  The same
    numbers wherever it runs -- the fixture generator (tests/golden/make_config3_golden.py) and the GPU tests feed on it."""
    sm = job['sm']
    E = job['num_expressions']
    rng = np.random.default_rng(motion_seed + 5)
    pose_gt, trans_gt = synth.synth_motion(sm.NP, sm.body_dof, n_frames, seed=motion_seed)
    t = np.arange(n_frames)[:, None] / 30.0
    pose_gt[:, 66:69] = 0.15 * np.sin(2 * np.pi * 1.1 * t + np.array([0.0, 1.0, 2.0]))   # jaw
    pose_gt[:, 69:75] = 0.0
    shp = expr_amp * rng.standard_normal(E)
    M = closest.shape[0]
    flat = closest.reshape(-1)
    markers = np.zeros((n_frames, M, 3))
    for f in range(n_frames):
        v = so.verts_forward(m, so.fullpose_from_pose(m, pose_gt[f]), trans_gt[f], flat, shp=shp).reshape(M, 3, 3)
        markers[f] = so.markers_from_verts(coef, v[:, 0], v[:, 1], v[:, 2])
    markers += rng.normal(0, noise, markers.shape)
    drop = rng.random(markers.shape[:2]) < dropout
    drop[0, :] = False
    markers[drop] = 0.0
    return dict(obs=markers, vis=~drop, pose_gt=pose_gt, trans_gt=trans_gt, expr_gt=shp)


def oracle_of_job(job):
    """The oracle's model / prior / attachment of a workload.make_job job (bench.py's oracle_setup; the parity criterion's marker check)."""
    sm = job['sm']
    m = so.prepare_model(dict(v_template=sm.v_template, shapedirs=sm.shapedirs, posedirs=sm.posedirs, weights=sm.weights,
                              J_regressor=sm.J_regressor, parents=sm.parents, body_dof=sm.body_dof, hand_dof=sm.hand_dof,
                              hands_mean=sm.hands_mean, selected_components=sm.selected_components), job['betas'])
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3))
    closest, coef = so.transformed_coeffs(can, job['markers_latent'])
    return m, job['prior'], closest, coef
