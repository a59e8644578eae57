"""Max-mixture GMM body-pose prior, host-side preparation.

Mirror of `create_gmm_body_prior` (reference src/moshpp/prior/gmm_prior_ch.py:107-134).  Evaluation
(`MaxMixtureComplete`, :42-85) runs inside the HIP chain kernel; this module only prepares the constants.
"""
from __future__ import annotations

import os
import pickle

import numpy as np


def create_gmm_body_prior(pose_body_prior_fname, exclude_hands=False):
    """-> dict(means[G,npose], chols[G,npose,npose], weights[G], npose).
    `pose_body_prior_fname` is the pickle path (keys 'means', 'covars', 'weights') or such a dict."""
    if isinstance(pose_body_prior_fname, dict):
        gmm = pose_body_prior_fname
    else:
        assert os.path.exists(pose_body_prior_fname), \
            ValueError(f'pose_body_prior_fname does not exist: {pose_body_prior_fname}')
        with open(pose_body_prior_fname, 'rb') as f:
            gmm = pickle.load(f, encoding='latin-1')
    npose = 63 if exclude_hands else 69
    covars = np.asarray(gmm['covars'], dtype=np.float64)[:, :npose, :npose]
    means = np.asarray(gmm['means'], dtype=np.float64)[:, :npose]
    weights = np.asarray(gmm['weights'], dtype=np.float64).ravel()
    precs = np.asarray([np.linalg.inv(cov) for cov in covars])
    chols = np.asarray([np.linalg.cholesky(prec) for prec in precs])
    sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in covars])
    const = (2 * np.pi) ** (npose / 2.)
    weights = weights / (const * (sqrdets / sqrdets.min()))
    return dict(means=np.ascontiguousarray(means), chols=np.ascontiguousarray(chols), weights=weights, npose=npose)
