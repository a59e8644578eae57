"""Marker attachment, host side.

Mirror of `TransformedCoeffs` (reference src/moshpp/transformed_lm.py:45-113) for the way Stage-II uses it:
numeric inputs, evaluated once per sequence (chmosh.py:502).  The per-evaluation half, `TransformedLms`
(:120-162), runs inside the HIP chain kernel (csrc/chain_solve.hip: marker_eval).
"""
from __future__ import annotations

import numpy as np

# support_data/smplx_eyeballs.npz holds vertex ids 9383..10474; the reference keeps
# set(arange(10474)) - eyeballs, i.e. vertices 0..9382, when the body has 10475 vertices (:48-50, 67-69).
SMPLX_NUM_VERTS = 10475
SMPLX_FIRST_EYEBALL_VID = 9383


def _nrm(x):
    with np.errstate(invalid='ignore', divide='ignore'):
        return x / np.sqrt(np.sum(x ** 2, axis=1)).reshape((-1, 1))


class TransformedCoeffs:
    """closest[M,3] (vertex ids into the full body) and coef[M,3] of every latent marker.

    8 nearest neighbours on the canonical body (eyeballs excluded for SMPL-X); local frame
    f1 = nrm(v1-v0), f2 = nrm(e1 x e2), f3 = f1 x f2; if ANY marker's cross product is degenerate the
    third neighbour is replaced by the next-nearest for ALL markers (:94-101)."""

    n_neighbors = 8

    def __init__(self, can_body, markers_latent):
        can_body = np.asarray(can_body, dtype=np.float64)
        markers_latent = np.asarray(markers_latent, dtype=np.float64).reshape(-1, 3)
        if len(can_body) == SMPLX_NUM_VERTS:
            keep = np.arange(SMPLX_FIRST_EYEBALL_VID)
        else:
            keep = np.arange(len(can_body))
        pts = can_body[keep]
        d2 = ((markers_latent[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
        closest = np.argsort(d2, axis=1, kind='stable')[:, :self.n_neighbors]
        diff = markers_latent - pts[closest[:, 0]]
        e1 = pts[closest[:, 1]] - pts[closest[:, 0]]
        e2 = pts[closest[:, 2]] - pts[closest[:, 0]]
        f1 = _nrm(e1)
        nn = 3
        while np.isnan(_nrm(np.cross(e1, e2)).sum()) and nn < closest.shape[0]:
            e2 = pts[closest[:, nn]] - pts[closest[:, 0]]
            nn += 1
        closest[:, 2] = closest[:, nn - 1]
        f2 = _nrm(np.cross(e1, e2))
        f3 = np.cross(f1, f2)
        self.closest = keep[closest[:, :3]].astype(np.int32)
        self.coef = np.stack([(diff * f1).sum(1), (diff * f2).sum(1), (diff * f3).sum(1)], axis=1)
        if not np.all(np.isfinite(self.coef)):
            raise ValueError('degenerate marker attachment (collinear neighbours for every candidate)')
