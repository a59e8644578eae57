"""Synthetic Stage-II jobs shaped like the BASELINE.json configs (bench.py, tests, smoke)."""
from __future__ import annotations

import numpy as np

from . import synth
from .cfg import STAGEII_WEIGHTS
from .models import load_surface_model
from .prior import create_gmm_body_prior

CONFIGS = {   # BASELINE.json "configs" (free variables / residual rows per frame)
    'config0_smpl_120f_41mk': dict(model_type='smpl', n_frames=120, n_markers=41, optimize_fingers=False),
    'config1_smplh_4000f_53mk': dict(model_type='smplh', n_frames=4000, n_markers=53, optimize_fingers=False),
    'config2_smplx_4000f_89mk': dict(model_type='smplx', n_frames=4000, n_markers=89, optimize_fingers=True),
    'config3_mano_10000f_33mk': dict(model_type='mano', n_frames=10000, n_markers=33, optimize_fingers=True),
}


def surface_model_from_synth(seq):
    """SurfaceModel through the same loader path a model file takes (models.load_surface_model)."""
    dd = {k: v for k, v in seq['model'].items() if not k.startswith('_')}
    return load_surface_model(dd, pose_hand_prior_fname=seq['hand_prior'], use_hands_mean=seq['use_hands_mean'],
                              dof_per_hand=seq['dof_per_hand'], surface_model_type=seq['model_type'])


def make_job(model_type='smplh', n_frames=4000, n_markers=53, seed=0, optimize_fingers=False, dd=None, **kw):
    """Host-side description of one sequence: SurfaceModel, prepared prior, betas, latent markers,
    obs[F,M,3] / vis[F,M] already ordered by latent label, Stage-II weights."""
    body_only = not optimize_fingers
    seq = synth.make_sequence(model_type, n_frames, n_markers, seed=seed, body_only_markers=body_only, dd=dd, **kw)
    sm = surface_model_from_synth(seq)
    prior = None
    if model_type != 'mano':
        prior = create_gmm_body_prior(seq['gmm'], exclude_hands=model_type in ('smplh', 'smplx'))
    obs = np.nan_to_num(seq['markers'])
    vis = ~np.isnan(seq['markers']).any(-1)
    return dict(seq=seq, sm=sm, prior=prior, betas=seq['betas'], markers_latent=seq['markers_latent'],
                obs=obs, vis=vis, weights=dict(STAGEII_WEIGHTS['smplh']), model_type=model_type,
                optimize_fingers=optimize_fingers)


def make_solver(job, maxiter=100):
    from .chmosh import StageIISolver
    return StageIISolver(job['sm'], job['betas'], job['markers_latent'], job['prior'], job['weights'],
                         surface_model_type=job['model_type'], optimize_fingers=job['optimize_fingers'], maxiter=maxiter)
