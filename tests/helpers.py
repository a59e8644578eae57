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


def device_case(case, optimize_fingers=False, optimize_toes=False, maxiter=100, weights=None):
    """libmoshii handles + options for an oracle_case (same arrays, same ids)."""
    from moshpp_amd import capi
    mdl = case['model']
    m = case['m']
    dev = capi.Model(mdl['v_template'], mdl['shapedirs'], mdl['posedirs'], mdl['weights'], mdl['J_regressor'],
                     mdl['parents'], mdl['body_dof'], mdl['hand_dof'], mdl['hands_mean'], mdl['selected_components'])
    dev.set_betas(case['s']['betas'])
    att = capi.Attachment(dev, case['closest'], case['coef'])
    pr = None
    if case['prior'] is not None:
        pr = capi.Prior(case['prior']['means'], case['prior']['chols'], case['prior']['weights'])
    root, body, finger, st1, st2 = so.pose_id_sets(case['model_type'], m['NP'], optimize_fingers, optimize_toes)
    W = so.stageii_weights_default() if weights is None else weights
    opts = capi.make_opts(W, st1, st2, body, finger if optimize_fingers else [], maxiter=maxiter)
    return dict(model=dev, attach=att, prior=pr, opts=opts)
