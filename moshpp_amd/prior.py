"""Max-mixture GMM body-pose prior, host-side preparation.

Mirror of `create_gmm_body_prior` (reference src/moshpp/prior/gmm_prior_ch.py:107-134).  Evaluation
(`MaxMixtureComplete`, :42-85) runs inside the HIP chain kernel; this module only prepares the constants.
"""
from __future__ import annotations

import os
import pickle

import numpy as np


def _load_mixture(source):
    if isinstance(source, dict):
        return source
    if not os.path.exists(source):
        raise AssertionError(ValueError(f'pose_body_prior_fname does not exist: {source}'))
    with open(source, 'rb') as fh:
        return pickle.load(fh, encoding='latin-1')


def create_gmm_body_prior(pose_body_prior_fname, exclude_hands=False):
    """-> dict(means[G,npose], chols[G,npose,npose], weights[G], npose).
    `pose_body_prior_fname` is the pickle path (keys 'means', 'covars', 'weights') or such a dict.

    Per component g the kernel needs a factor L_g with L_g L_g^T = Sigma_g^-1 (the residual is L_g^T (x - mu_g)) and the mixture
    weight divided by the Gaussian's normalisation, (2 pi)^(npose/2) sqrt(det Sigma_g), the determinants taken relative to the
    smallest one (gmm_prior_ch.py:121-131).  Both come out of ONE Cholesky factorisation of the covariance block here,
    Sigma_g = C C^T:  det Sigma_g = prod(diag C)^2, Sigma_g^-1 = C^-T C^-1 -- whose lower Cholesky factor (unique: positive diagonal)
    is obtained by factorising the explicitly symmetrised inverse."""
    mix = _load_mixture(pose_body_prior_fname)
    npose = 63 if exclude_hands else 69
    mu = np.ascontiguousarray(np.asarray(mix['means'], dtype=np.float64)[:, :npose])
    sigma = np.asarray(mix['covars'], dtype=np.float64)[:, :npose, :npose]
    pi_g = np.asarray(mix['weights'], dtype=np.float64).ravel()
    factors = np.empty_like(sigma)
    root_det = np.empty(len(sigma))
    for g, cov in enumerate(sigma):
        root_det[g] = np.sqrt(np.linalg.det(cov))
        factors[g] = np.linalg.cholesky(np.linalg.inv(cov))
    norm = (2.0 * np.pi) ** (0.5 * npose) * (root_det / root_det.min())
    return dict(means=mu, chols=np.ascontiguousarray(factors), weights=pi_g / norm, npose=npose)
