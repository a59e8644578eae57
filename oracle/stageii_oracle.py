"""TEST INFRASTRUCTURE ONLY -- float64 NumPy restatement of MoSh++ Stage-II.

*** PARITY PARTLY PINNED ***  The reference (nghorbani/moshpp v3.0) delegates the
arithmetic of this path to third-party packages that are neither vendored in
/root/reference nor installable in this environment (no network):

  * chumpy (unpinned, requirements.txt:2)      -- autodiff graph + `minimize_dogleg`
  * psbody.smpl (not listed in requirements)   -- `verts_decorated`, C++ `lbs_derivatives_wrt_pose`
  * cv2.Rodrigues, sklearn KD-tree

and the reference has no tests, golden vectors or fixtures.

PINNED to the reference's own code, executed in the build container (fixtures + generating
scripts under tests/golden/, checked by tests/test_ref_golden.py and tests/test_host_logic.py):
  * transformed_coeffs / markers_from_verts  <- TransformedCoeffs / TransformedLms values (transformed_lm.py:45-162,
    incl. the sklearn kd-tree 8-NN and the SMPL-X eyeball exclusion)
  * prepare_gmm_prior / gmm_prior_eval       <- create_gmm_body_prior + MaxMixtureComplete values (gmm_prior_ch.py:42-134)
  * rigid_landmark_transform                 <- rigid_transformations.py:39-69
  * (host package) C3D reader / writer       <- the reference's vendored py-c3d writer and reader (tools/c3d.py)
  * (host package) MocapSession, load_surface_model, AMASS part split <- the reference's own class / functions
    (tools/mocap_interface.py:87-279, models/smpl_fast_derivatives.py:52-150, tools/run_tools.py:70-85)
  * stageii_chain (the Stage-II schedule: first-frame rigid init and annealed rounds, Step 1 / Step 2 free sets, weights,
    velocity target, empty frames, result terms) and the way `minimize_dogleg` is driven  <- the reference's own
    `mosh_stageii` function (chmosh.py:468-741) EXECUTED under a lazy chumpy stand-in with the reference's node classes
    (tests/golden/make_ref_stageii_golden.py -> tests/golden/ref_stageii.npz; 7 cases: body, SMPL, fingers, toes, MANO, face + expressions, DMPL): <= 5e-9 rad, equal
    dogleg iteration counts on every solve
UNPINNED (third-party code absent): the LBS arithmetic of psbody.smpl (forward + pose Jacobian) and the internals of
chumpy's `minimize_dogleg` (radius rules, stops, the solve).  These restate the *published* algorithms (SMPL's public
`lbs.py`/`posemapper.py`/`verts.py`, chumpy's `optimization_internal.py`), anchored on the reference's call sites and
validated by internal self-checks only (finite differences, scipy least-squares minimum, ground-truth recovery) --
see tests/test_oracle.py.

Reference anchors (all relative to /root/reference/src/moshpp):
  chmosh.py:458-741                    Stage-II schedule, weights, free variables, outputs
  models/smpl_fast_derivatives.py:169-263   SmplModelLBS forward, hand-PCA map, Jacobian chaining
  transformed_lm.py:45-162             marker attachment (TransformedCoeffs / TransformedLms)
  prior/gmm_prior_ch.py:42-134         max-mixture GMM pose prior
  rigid_transformations.py:39-83       first-frame rigid initialisation

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    'rodrigues', 'rotmat_to_rotvec', 'prepare_model', 'fullpose_from_pose', 'joint_transforms',
    'verts_forward', 'verts_jacobian', 'markers_from_verts', 'transformed_coeffs',
    'prepare_gmm_prior', 'gmm_prior_eval', 'rigid_landmark_transform', 'minimize_dogleg',
    'StageIIObjective', 'stageii_chain', 'stageii_weights_default',
]

_SMALL_T2 = 1e-6  # |theta|^2 below which the series forms are used (shared with the HIP kernels)


# ------------------------------------------------------------------------------------------
# Rodrigues (chumpy's Rodrigues node wraps cv2.Rodrigues)
# ------------------------------------------------------------------------------------------
def _skew(v):
    v = np.asarray(v, dtype=np.float64)
    out = np.zeros(v.shape[:-1] + (3, 3))
    out[..., 0, 1] = -v[..., 2]
    out[..., 0, 2] = v[..., 1]
    out[..., 1, 0] = v[..., 2]
    out[..., 1, 2] = -v[..., 0]
    out[..., 2, 0] = -v[..., 1]
    out[..., 2, 1] = v[..., 0]
    return out


def rodrigues(r):
    """Axis-angle (...,3) -> rotation R (...,3,3) and SO(3) left Jacobian Jl (...,3,3).

    R = I + a K + b K^2,  Jl = I + b K + c K^2,  K = [r]x,
    a = sin t / t, b = (1-cos t)/t^2, c = (t - sin t)/t^3 (series below t^2 < 1e-6).
    dR/dr_c = [Jl[:, c]]x R   (equals cv2.Rodrigues' 9x3 Jacobian analytically).
    """
    r = np.asarray(r, dtype=np.float64)
    t2 = np.sum(r * r, axis=-1)
    t = np.sqrt(t2)
    small = t2 < _SMALL_T2
    ts = np.where(small, 1.0, t)
    t2s = np.where(small, 1.0, t2)
    a = np.where(small, 1.0 - t2 / 6.0 + t2 * t2 / 120.0, np.sin(ts) / ts)
    b = np.where(small, 0.5 - t2 / 24.0 + t2 * t2 / 720.0, (1.0 - np.cos(ts)) / t2s)
    c = np.where(small, 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0, (ts - np.sin(ts)) / (t2s * ts))
    K = _skew(r)
    K2 = K @ K
    eye = np.eye(3)
    R = eye + a[..., None, None] * K + b[..., None, None] * K2
    Jl = eye + b[..., None, None] * K + c[..., None, None] * K2
    return R, Jl


def rotmat_to_rotvec(R):
    """cv2.Rodrigues(R)[0] for a proper rotation: axis-angle with angle in [0, pi]
    (rigid_transformations.py:82)."""
    R = np.asarray(R, dtype=np.float64)
    rx = R[2, 1] - R[1, 2]
    ry = R[0, 2] - R[2, 0]
    rz = R[1, 0] - R[0, 1]
    s = np.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = (R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5
    c = min(1.0, max(-1.0, c))
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        # angle ~ pi: recover the axis from the symmetric part
        t = (R[0, 0] + 1.0) * 0.5
        x = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1.0) * 0.5
        y = np.sqrt(max(t, 0.0)) * (1.0 if R[0, 1] >= 0 else -1.0)
        t = (R[2, 2] + 1.0) * 0.5
        z = np.sqrt(max(t, 0.0)) * (1.0 if R[0, 2] >= 0 else -1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and (R[1, 2] > 0) != (y * z > 0):
            z = -z
        v = np.array([x, y, z])
        return v * (theta / np.linalg.norm(v))
    return np.array([rx, ry, rz]) * (0.5 * theta / s)


# ------------------------------------------------------------------------------------------
# Body model (smpl_fast_derivatives.py:169-244 + public SMPL lbs/posemapper/verts, Appendix A.1)
# ------------------------------------------------------------------------------------------
def prepare_model(model, betas=None):
    """Freeze shape: v_shaped = v_template + shapedirs[:,:,:nb].betas (smpl_fast_derivatives.py:186),
    J = J_regressor.v_shaped (:187-191).  `model` is a dict of plain arrays:
    v_template[V,3] shapedirs[V,3,NB] posedirs[V,3,9(K-1)] weights[V,K] J_regressor[K,V]
    parents[K] body_dof hand_dof hands_mean[P-body_dof] selected_components[hand_dof,P-body_dof].
    Returns a new dict with v_shaped, J, ancestors matrix and f64 copies."""
    m = dict(model)
    for k in ('v_template', 'shapedirs', 'posedirs', 'weights', 'hands_mean', 'selected_components'):
        if k in m and m[k] is not None:
            m[k] = np.asarray(m[k], dtype=np.float64)
    Jreg = m['J_regressor']
    if hasattr(Jreg, 'toarray'):
        Jreg = Jreg.toarray()
    m['J_regressor'] = np.asarray(Jreg, dtype=np.float64)
    parents = np.asarray(m['parents'], dtype=np.int64)
    m['parents'] = parents
    K = len(parents)
    nb_tot = m['shapedirs'].shape[2]
    b = np.zeros(nb_tot)
    if betas is not None:
        betas = np.asarray(betas, dtype=np.float64).ravel()
        b[:min(len(betas), nb_tot)] = betas[:nb_tot]
    m['betas'] = b
    m['v_shaped'] = m['v_template'] + m['shapedirs'].dot(b)
    m['J'] = m['J_regressor'].dot(m['v_shaped'])
    anc = np.zeros((K, K))  # anc[k, j] = 1 iff k is j or an ancestor of j
    for j in range(K):
        a = j
        while a >= 0:
            anc[a, j] = 1.0
            a = parents[a]
    m['anc'] = anc
    m['K'] = K
    m['P'] = 3 * K
    m['body_dof'] = int(m['body_dof'])
    m['hand_dof'] = int(m.get('hand_dof', 0) or 0)
    m['NP'] = m['body_dof'] + m['hand_dof']
    return m


def set_free_shape(m, start, count):
    """Declare shapedirs columns [start, start+count) as per-frame free variables of Stage-II Step 2: the
    expression coefficients `opt_model.betas[exp_start:exp_start+num_expressions]` (chmosh.py:565-567, 687-688) or the
    DMPL coefficients `opt_model.betas[num_betas:num_betas+num_dmpls]` (:513-514, 698-699).  The coefficients act as an
    OFFSET on top of the betas frozen by prepare_model.  Both v_shaped and the regressed joints move with them
    (smpl_fast_derivatives.py:186-191): S_free[V,3,E] and JS = J_regressor . S_free [K,3,E]."""
    m['shape_start'], m['E'] = int(start), int(count)
    m['S_free'] = np.ascontiguousarray(m['shapedirs'][:, :, start:start + count])
    m['JS'] = np.einsum('kv,vie->kie', m['J_regressor'], m['S_free'])
    return m


def fullpose_from_pose(m, pose):
    """smpl_fast_derivatives.py:194-204: fullpose = [pose[:body_dof], hands_mean + pose_hand . comps]."""
    pose = np.asarray(pose, dtype=np.float64)
    bd, hd = m['body_dof'], m['hand_dof']
    if hd == 0:
        return pose[:bd].copy()
    hand = m['hands_mean'] + pose[bd:bd + hd].dot(m['selected_components'])
    return np.concatenate([pose[:bd], hand])


def pose_map_matrix(m):
    """d fullpose / d pose  (P x NP): the `m` matrix of smpl_fast_derivatives.py:250-254."""
    bd, hd, P = m['body_dof'], m['hand_dof'], m['P']
    mm = np.zeros((P, bd + hd))
    mm[:bd, :bd] = np.eye(bd)
    if hd:
        mm[bd:, bd:] = m['selected_components'].T
    return mm


def joint_transforms(m, fullpose, J=None):
    """global_rigid_transformation (Appendix A.1): world rotations Rw[K,3,3], joint world
    positions tw[K,3] (excluding trans); also local R, left Jacobians Jl.  `J` overrides the frozen joints."""
    K, parents = m['K'], m['parents']
    J = m['J'] if J is None else J
    R, Jl = rodrigues(fullpose.reshape(K, 3))
    Rw = np.zeros((K, 3, 3))
    tw = np.zeros((K, 3))
    Rw[0] = R[0]
    tw[0] = J[0]
    for j in range(1, K):
        p = parents[j]
        Rw[j] = Rw[p].dot(R[j])
        tw[j] = Rw[p].dot(J[j] - J[p]) + tw[p]
    return R, Jl, Rw, tw


def _shaped(m, sl, shp):
    """v_shaped rows and regressed joints at the free shape coefficients `shp` (None: the frozen ones)."""
    if shp is None:
        return m['v_shaped'][sl], m['J']
    shp = np.asarray(shp, dtype=np.float64)
    return m['v_shaped'][sl] + m['S_free'][sl].dot(shp), m['J'] + m['JS'].dot(shp)


def verts_forward(m, fullpose, trans, vids=None, shp=None):
    """LBS forward for the vertex subset `vids` (all if None):
    v = sum_j w_vj (Rw_j (v_posed - J_j) + tw_j) + trans, v_posed = v_shaped + posedirs.vec(R_j - I)."""
    sl = slice(None) if vids is None else vids
    v_shaped, J = _shaped(m, sl, shp)
    R, Jl, Rw, tw = joint_transforms(m, fullpose, J)
    feat = (R[1:] - np.eye(3)).reshape(-1)
    v_posed = v_shaped + m['posedirs'][sl].dot(feat)
    w = m['weights'][sl]
    # T = sum_j w_j A_j, A_j = [Rw_j | tw_j - Rw_j J_j]
    Arot = Rw
    Atr = tw - np.einsum('kab,kb->ka', Rw, J)
    Trot = np.einsum('nk,kab->nab', w, Arot)
    Ttr = w.dot(Atr)
    v = np.einsum('nab,nb->na', Trot, v_posed) + Ttr + np.asarray(trans, dtype=np.float64)
    return v


def verts_jacobian(m, fullpose, trans, vids, shp=None, want_shape=False):
    """Value and analytic Jacobian of the vertex subset wrt every fullpose dof (and, with want_shape, wrt the
    free shape coefficients: psbody's `lbs_derivatives_wrt_shape`, smpl_fast_derivatives.py:260-261, restated).

    Returns v[n,3], dv[n,3,P].  For dof (k,c):
      dv = omega_kc x (S_k - W_k tw_k)  +  Trot . (posedirs[v,:,9(k-1):9k] . vec(dR_k/dtheta_c))   (k>=1 for 2nd term)
    with omega_kc = Rw_par(k) Jl_k[:,c], S_k = sum_{j in subtree(k)} w_j x_j, x_j = Rw_j (v_posed-J_j)+tw_j,
    W_k = sum_{j in subtree(k)} w_j.  d/dtrans = I (not returned).

    Shape: with s the coefficients, dv_posed/ds_e = S_e(v), dJ_j/ds_e = JS_je, and the joint world positions obey
    dt_0 = JS_0, dt_j = dt_par + Rw_par (JS_j - JS_par); hence
      dv/ds_e = Trot . S_e(v) + sum_j w_j q_je,   q_je = dt_je - Rw_j JS_je."""
    K, parents = m['K'], m['parents']
    v_shaped, J = _shaped(m, vids, shp)
    R, Jl, Rw, tw = joint_transforms(m, fullpose, J)
    feat = (R[1:] - np.eye(3)).reshape(-1)
    Pd = m['posedirs'][vids]  # n,3,9(K-1)
    n = Pd.shape[0]
    v_posed = v_shaped + Pd.dot(feat)
    w = m['weights'][vids]
    x = np.einsum('kab,nkb->nka', Rw, v_posed[:, None, :] - J[None]) + tw[None]  # n,K,3
    v = np.einsum('nk,nka->na', w, x) + np.asarray(trans, dtype=np.float64)
    Trot = np.einsum('nk,kab->nab', w, Rw)
    anc = m['anc']
    S = np.einsum('kj,nj,nja->nka', anc, w, x)
    Wk = np.einsum('kj,nj->nk', anc, w)
    arm = S - Wk[..., None] * tw[None]  # n,K,3
    # world rotation axes
    Rpar = np.empty((K, 3, 3))
    Rpar[0] = np.eye(3)
    Rpar[1:] = Rw[parents[1:]]
    omega = np.einsum('kab,kbc->kca', Rpar, Jl)  # omega[k,c,:] = Rpar_k @ Jl_k[:,c]
    dv_art = np.cross(omega[None, :, :, :], arm[:, :, None, :])  # n,K,3(c),3(xyz)
    # pose-corrective part: B[k,c] = skew(Jl_k[:,c]) @ R_k
    B = np.einsum('kcab,kbd->kcad', _skew(np.swapaxes(Jl, 1, 2)), R)  # K,3(c),3,3
    Bf = B[1:].reshape(K - 1, 3, 9)
    Pd4 = Pd.reshape(n, 3, K - 1, 9)
    pd = np.einsum('nike,kce->nkci', Pd4, Bf)  # n,K-1,3(c),3(i)
    dv_cor = np.einsum('nab,nkcb->nkca', Trot, pd)
    dv = dv_art
    dv[:, 1:] += dv_cor
    dv = np.transpose(dv, (0, 3, 1, 2)).reshape(n, 3, 3 * K)
    if not want_shape:
        return v, dv
    JS = m['JS']  # K,3,E
    dt = np.zeros_like(JS)
    dt[0] = JS[0]
    for j in range(1, K):
        p = parents[j]
        dt[j] = dt[p] + Rw[p].dot(JS[j] - JS[p])
    q = dt - np.einsum('kab,kbe->kae', Rw, JS)
    dv_shape = np.einsum('nab,nbe->nae', Trot, m['S_free'][vids]) + np.einsum('nk,kae->nae', w, q)
    return v, dv, dv_shape


def verts_jacobian_reference_cost(m, fullpose, trans, vids):
    """Same numbers as verts_jacobian, computed the way the reference pays for them:
    full-mesh forward and the dense 3V x 3K pose Jacobian (psbody C++ `lbs_derivatives_wrt_pose`),
    then row selection (smpl_fast_derivatives.py:246-258).  CPU-baseline timing only."""
    allv = np.arange(m['v_shaped'].shape[0])
    v, dv = verts_jacobian(m, fullpose, trans, allv)
    return v[vids], dv[vids]


# ------------------------------------------------------------------------------------------
# Marker attachment (transformed_lm.py)
# ------------------------------------------------------------------------------------------
def _nrm(x):
    with np.errstate(invalid='ignore', divide='ignore'):
        return x / np.sqrt(np.sum(x ** 2, axis=1)).reshape((-1, 1))


def transformed_coeffs(can_body, markers_latent, exclude_vids=None, n_neighbors=8):
    """TransformedCoeffs.on_changed (transformed_lm.py:59-113).  8-NN of each latent marker on the
    canonical body (eyeball vertices excluded for V=10475, :49-50,67-74); local frame from the
    3 nearest; collinearity fallback swaps the 3rd neighbour for ALL markers (:94-101).
    Returns closest[M,3] (global vertex ids) and coef[M,3]."""
    can_body = np.asarray(can_body, dtype=np.float64)
    markers_latent = np.asarray(markers_latent, dtype=np.float64)
    V = can_body.shape[0]
    keep = np.arange(V)
    if exclude_vids is not None and len(exclude_vids):
        mask = np.ones(V, dtype=bool)
        mask[np.asarray(exclude_vids, dtype=np.int64)] = False
        keep = keep[mask]
    pts = can_body[keep]
    d2 = ((markers_latent[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    order = np.argsort(d2, axis=1, kind='stable')[:, :n_neighbors]
    closest = order.copy()
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
    coef = np.stack([(diff * f1).sum(1), (diff * f2).sum(1), (diff * f3).sum(1)], axis=1)
    return keep[closest[:, :3]], coef


def markers_from_verts(coef, v0, v1, v2, want_jac=False):
    """TransformedLms.on_changed (transformed_lm.py:130-162):
    m = v0 + c0 f1 + c1 f2 + c2 f3 with f1 = nrm(v1-v0), f2 = nrm(e1 x e2), f3 = f1 x f2.
    Optionally the 3x9 Jacobian wrt (v0, v1, v2) per marker."""
    e1 = v1 - v0
    e2 = v2 - v0
    l1 = np.sqrt((e1 * e1).sum(1))[:, None]
    f1 = e1 / l1
    nv = np.cross(e1, e2)
    ln = np.sqrt((nv * nv).sum(1))[:, None]
    f2 = nv / ln
    f3 = np.cross(f1, f2)
    c = coef
    mk = v0 + c[:, 0:1] * f1 + c[:, 1:2] * f2 + c[:, 2:3] * f3
    if not want_jac:
        return mk
    M = v0.shape[0]
    eye = np.eye(3)
    # df1/de1, df2/dn
    D1 = (eye[None] - f1[:, :, None] * f1[:, None, :]) / l1[:, :, None]
    D2 = (eye[None] - f2[:, :, None] * f2[:, None, :]) / ln[:, :, None]
    # dn/de1 = -[e2]x ; dn/de2 = [e1]x
    dn_de1 = -_skew(e2)
    dn_de2 = _skew(e1)
    df2_de1 = D2 @ dn_de1
    df2_de2 = D2 @ dn_de2
    # f3 = f1 x f2: df3 = -[f2]x df1 + [f1]x df2
    sf1 = _skew(f1)
    sf2 = _skew(f2)
    df3_de1 = -sf2 @ D1 + sf1 @ df2_de1
    df3_de2 = sf1 @ df2_de2
    c0 = c[:, 0][:, None, None]
    c1 = c[:, 1][:, None, None]
    c2 = c[:, 2][:, None, None]
    dm_de1 = c0 * D1 + c1 * df2_de1 + c2 * df3_de1
    dm_de2 = c1 * df2_de2 + c2 * df3_de2
    L = np.zeros((M, 3, 9))
    L[:, :, 0:3] = eye[None] - dm_de1 - dm_de2
    L[:, :, 3:6] = dm_de1
    L[:, :, 6:9] = dm_de2
    return mk, L


# ------------------------------------------------------------------------------------------
# GMM max-mixture prior (prior/gmm_prior_ch.py)
# ------------------------------------------------------------------------------------------
def prepare_gmm_prior(gmm, npose):
    """create_gmm_body_prior (gmm_prior_ch.py:107-134): chols = chol(inv(cov)),
    weights /= (2pi)^(npose/2) * sqrt(det)/min(sqrt(det))."""
    covars = np.asarray(gmm['covars'], dtype=np.float64)[:, :npose, :npose]
    means = np.asarray(gmm['means'], dtype=np.float64)[:, :npose]
    weights = np.asarray(gmm['weights'], dtype=np.float64).ravel()
    precs = np.asarray([np.linalg.inv(cov) for cov in covars])
    chols = np.asarray([np.linalg.cholesky(prec) for prec in precs])
    sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in covars])
    const = (2 * np.pi) ** (npose / 2.)
    weights = weights / (const * (sqrdets / sqrdets.min()))
    return {'means': means, 'chols': chols, 'weights': weights, 'npose': npose}


def gmm_prior_eval(prior, x, want_jac=False):
    """MaxMixtureComplete (gmm_prior_ch.py:53-85): l_k = sqrt(.5) (x-mu_k).L_k; k* = argmin(|l_k|^2 - log w_k);
    residual [l_k*, sqrt(-log w_k*)] (npose+1); Jacobian sqrt(.5) L_k*^T with a zero last row."""
    means, chols, weights = prior['means'], prior['chols'], prior['weights']
    ll = np.sqrt(0.5) * np.einsum('gb,gba->ga', x[None, :] - means, chols)
    score = (ll ** 2).sum(1) - np.log(weights)
    k = int(np.argmin(score))
    r = np.concatenate([ll[k], [np.sqrt(-np.log(weights[k]))]])
    if not want_jac:
        return r, k
    Jp = np.zeros((len(x) + 1, len(x)))
    Jp[:-1] = np.sqrt(0.5) * chols[k].T
    return r, k, Jp


# ------------------------------------------------------------------------------------------
# Rigid initialisation (rigid_transformations.py:39-83)
# ------------------------------------------------------------------------------------------
def rigid_landmark_transform(a, b):
    """Arun et al.: (R, T) with R a + T ~= b; a, b are 3xN."""
    b = np.where(np.isnan(b), a, b)
    a_mean = np.mean(a, axis=1).reshape((-1, 1))
    b_mean = np.mean(b, axis=1).reshape((-1, 1))
    c = (a - a_mean).dot((b - b_mean).T)
    u, s, v = np.linalg.svd(c, full_matrices=False)
    v = v.T
    R = v.dot(u.T)
    if np.linalg.det(R) < 0:
        v[:, 2] = -v[:, 2]
        R = v.dot(u.T)
    T = (b_mean - R.dot(a_mean)).reshape((-1, 1))
    return R, T


# ------------------------------------------------------------------------------------------
# chumpy minimize_dogleg restated -- [EXT-RECALL]
# ------------------------------------------------------------------------------------------
def minimize_dogleg(obj, x0, e_3=0.0, delta_0=None, maxiter=200, e_1=1e-15, e_2=1e-15, stats=None):
    """Powell dogleg on sum r^2.  `obj` offers r(x) -> residual vector and J(x) -> dense Jacobian.
    Control flow follows chumpy.optimization_internal.minimize_dogleg / DoglegState:
      A = J^T J, g = J^T(-r); d_sd = |g|^2/|Jg|^2 g; stunted Cauchy if |d_sd| >= delta, else GN step
      (solved once per outer iteration) if inside the region, else the dogleg point;
      rho = (|r|^2-|r_new|^2) / (2 g.d - d.A.d) (divided only when the numerator > 0); accept iff rho > 0;
      after an accepted step stop if the relative improvement < e_3, else recompute J;
      radius: rho > .9 -> max(delta, 2.5|d|); rho < .05 -> delta/4; maxiter counts outer iterations.
    Returns x (1-D)."""
    p = np.asarray(x0, dtype=np.float64).copy()
    n_fev = 0
    n_jev = 0

    r = obj.r(p); n_fev += 1
    J = obj.J(p); n_jev += 1
    A = J.T.dot(J)
    g = J.T.dot(-r)
    delta = delta_0
    done = False
    iteration = 0
    if np.linalg.norm(g, np.inf) < e_1:
        done = True
    while not done:
        iteration += 1
        Jg = J.dot(g)
        d_sd = (np.linalg.norm(g) ** 2 / np.linalg.norm(Jg) ** 2) * g
        d_gn = None
        while True:
            # update_step
            if delta is not None and np.linalg.norm(d_sd) >= delta:
                d_dl = delta / np.linalg.norm(d_sd) * d_sd
            else:
                if d_gn is None:
                    try:
                        d_gn = np.linalg.solve(A, g)
                    except np.linalg.LinAlgError:
                        d_gn = np.linalg.lstsq(A, g, rcond=None)[0]
                if delta is None or np.linalg.norm(d_gn) <= delta:
                    d_dl = d_gn.copy()
                    if delta is None:
                        delta = np.linalg.norm(d_gn)
                else:
                    delta_sq = delta ** 2
                    diff = d_gn - d_sd
                    sqnorm_sd = np.linalg.norm(d_sd) ** 2
                    pnow = diff.dot(diff) * delta_sq + d_gn.dot(d_sd) ** 2 - np.linalg.norm(d_gn) ** 2 * sqnorm_sd
                    beta = (delta_sq - sqnorm_sd) / (diff.dot(d_sd) + np.sqrt(pnow))
                    d_dl = d_sd + beta * diff
            step_size = np.linalg.norm(d_dl)
            improved = False
            if step_size <= e_2 * np.linalg.norm(p):
                done = True
            else:
                r_new = obj.r(p + d_dl); n_fev += 1
                sse = r.dot(r)
                sse_new = r_new.dot(r_new)
                rho = sse - sse_new
                if rho > 0:
                    with np.errstate(divide='ignore', invalid='ignore'):
                        rho = rho / (2.0 * g.dot(d_dl) - d_dl.dot(A.dot(d_dl)))
                improved = bool(rho > 0)
                if improved:
                    p = p + d_dl
                    if e_3 > 0.0 and (sse - sse_new) / sse < e_3:
                        done = True
                    else:
                        J = obj.J(p); n_jev += 1
                        A = J.T.dot(J)
                        r = r_new
                        g = J.T.dot(-r)
                        if np.linalg.norm(g, np.inf) < e_1:
                            done = True
                # updateRadius
                if rho > 0.9:
                    delta = max(delta, 2.5 * np.linalg.norm(d_dl))
                elif rho < 0.05:
                    delta *= 0.25
                if delta <= e_2 * np.linalg.norm(p):
                    done = True
            if done or improved:
                break
        if iteration >= maxiter:
            done = True
    if stats is not None:
        stats['iterations'] = stats.get('iterations', 0) + iteration
        stats.setdefault('per_call', []).append(iteration)
        stats['fevals'] = stats.get('fevals', 0) + n_fev
        stats['jevals'] = stats.get('jevals', 0) + n_jev
    return p


# ------------------------------------------------------------------------------------------
# Stage-II objective + frame chain (chmosh.py:458-741)
# ------------------------------------------------------------------------------------------
def stageii_weights_default():
    """support_data/conf/moshpp_conf.yaml:118-125 (smplh / smplx tables)."""
    return dict(stageii_wt_data=400., stageii_wt_velo=2.5, stageii_wt_dmpl=1.0, stageii_wt_expr=1.0,
                stageii_wt_poseB=1.6, stageii_wt_poseH=1.0, stageii_wt_poseF=1.0, stageii_wt_annealing=2.5)


def face_pose_ids(model_type, optimize_face):
    """pose_face_ids (chmosh.py:560-563): the jaw, pose ids 66:69 of SMPL-X, when optimize_face."""
    return list(range(66, 69)) if (model_type == 'smplx' and optimize_face) else []


def pose_id_sets(model_type, NP, optimize_fingers=False, optimize_toes=False, optimize_face=False):
    """chmosh.py:546-579, 645-647, 665-667, 676-692.  Returns (root, body, finger, step1_ids, step2_ids)."""
    allp = list(range(NP))
    root = allp[:3]
    body, finger = [], []
    if model_type == 'smpl':
        body = allp[3:]
    elif model_type == 'smplh':
        body = allp[3:66]
        if optimize_fingers:
            finger = allp[66:]
    elif model_type == 'smplx':
        body = allp[3:66]
        if optimize_fingers:
            finger = allp[75:]
    elif model_type == 'mano':
        finger = allp[3:]
    else:
        raise ValueError(model_type)
    step1 = root + body
    if len(body) and not optimize_toes:
        step1 = sorted(set(step1).difference(set(allp[30:36])))
    step2 = list(step1)
    if optimize_fingers:
        step2 = step2 + finger
    step2 = step2 + face_pose_ids(model_type, optimize_face)   # :685-689
    step2 = sorted(set(step2))
    return root, body, finger, step1, step2


class StageIIObjective:
    """One frame's residual dict (chmosh.py:612-626, 681-699) over x = [trans, pose[free_ids], shape (if free)]
    (the ChInputsStacked view: chmosh.py:649, 668, 692; the reference stacks [trans, v_face_exp, pose[ids]] -- the
    column order only permutes the normal equations)."""

    def __init__(self, m, closest, coef, prior, body_ids, reference_cost=False):
        self.m = m
        self.closest = closest
        self.coef = coef
        self.prior = prior
        self.body_ids = np.asarray(body_ids, dtype=np.int64)
        self.reference_cost = reference_cost
        self.mm = pose_map_matrix(m)
        self.pose = np.zeros(m['NP'])
        self.trans = np.zeros(3)
        # per-frame terms
        self.vis = None
        self.obs = None
        self.wt_data = 0.0
        self.wt_pose = 0.0
        self.velo_target = None
        self.wt_velo = 0.0
        self.finger_ids = None
        self.wt_poseH = 0.0
        self.free_ids = None
        # Step-2 extras: jaw pose term, free shape coefficients (expression or DMPL) with their regulariser and the
        # "stay" term (chmosh.py:685-699)
        self.face_ids = None
        self.wt_poseF = 0.0
        self.shp = np.zeros(m['E']) if 'E' in m else None
        self.shape_free = False
        self.wt_shape = 0.0
        self.shp_anchor = None
        self.wt_stay = 0.0

    # -- state ----------------------------------------------------------------------------
    def x(self):
        parts = [self.trans, self.pose[self.free_ids]]
        if self.shape_free:
            parts.append(self.shp)
        return np.concatenate(parts)

    def set_x(self, x):
        nf = len(self.free_ids)
        self.trans = np.array(x[:3])
        self.pose[self.free_ids] = x[3:3 + nf]
        if self.shape_free:
            self.shp = np.array(x[3 + nf:])

    def _unpack(self, x):
        nf = len(self.free_ids)
        pose = self.pose.copy()
        pose[self.free_ids] = x[3:3 + nf]
        self._shp_eval = np.asarray(x[3 + nf:]) if self.shape_free else self.shp
        return pose, np.asarray(x[:3])

    # -- pieces ---------------------------------------------------------------------------
    def markers_sim(self, pose=None, trans=None, shp=None):
        pose = self.pose if pose is None else pose
        trans = self.trans if trans is None else trans
        shp = self.shp if shp is None else shp
        fp = fullpose_from_pose(self.m, pose)
        vids = self.closest.reshape(-1)
        v = verts_forward(self.m, fp, trans, vids, shp=shp).reshape(-1, 3, 3)
        return markers_from_verts(self.coef, v[:, 0], v[:, 1], v[:, 2])

    def terms(self, pose, trans, shp=None):
        """Ordered dict of residual blocks at (pose, trans, shape)."""
        out = {}
        shp = self.shp if shp is None else shp
        sim = self.markers_sim(pose, trans, shp)
        out['data'] = ((sim[self.vis] - self.obs[self.vis]) * self.wt_data).ravel()
        if len(self.body_ids):
            fp = fullpose_from_pose(self.m, pose)  # pose[body ids] are identity-mapped dofs
            rb, _ = gmm_prior_eval(self.prior, pose[self.body_ids])
            out['poseB'] = rb * self.wt_pose
            del fp
        if self.velo_target is not None:
            out['velo'] = (pose - self.velo_target) * self.wt_velo
        if self.finger_ids is not None:
            out['poseH'] = pose[self.finger_ids] * self.wt_poseH
        if self.face_ids is not None:
            out['poseF'] = pose[self.face_ids] * self.wt_poseF
        if self.shape_free:
            if self.shp_anchor is not None and self.wt_stay != 0.0:
                out['shape_stay'] = (shp - self.shp_anchor) * self.wt_stay
            out['shape'] = shp * self.wt_shape
        return out

    def r(self, x):
        pose, trans = self._unpack(x)
        return np.concatenate(list(self.terms(pose, trans, self._shp_eval).values()))

    def J(self, x):
        pose, trans = self._unpack(x)
        m = self.m
        NP = m['NP']
        free = np.asarray(self.free_ids, dtype=np.int64)
        fp = fullpose_from_pose(m, pose)
        vids = self.closest.reshape(-1)
        shp = self._shp_eval
        E = len(shp) if self.shape_free else 0
        dv_shape = None
        if self.reference_cost:
            v, dv = verts_jacobian_reference_cost(m, fp, trans, vids)
            dv_pose = np.matmul(dv.reshape(-1, m['P']), self.mm).reshape(len(vids), 3, NP)
        elif E:
            v, dv, dv_shape = verts_jacobian(m, fp, trans, vids, shp=shp, want_shape=True)
            dv_pose = dv.dot(self.mm)
        else:
            v, dv = verts_jacobian(m, fp, trans, vids, shp=shp)
            dv_pose = dv.dot(self.mm)  # n,3,NP
        M = self.closest.shape[0]
        v = v.reshape(M, 3, 3)
        _, L = markers_from_verts(self.coef, v[:, 0], v[:, 1], v[:, 2], want_jac=True)
        dvp = dv_pose.reshape(M, 9, NP)
        dm_pose = np.einsum('mab,mbp->map', L, dvp)  # M,3,NP
        vis = self.vis
        nobs = int(vis.sum())
        blocks = []
        nf = len(free)
        Jd = np.zeros((3 * nobs, 3 + nf + E))
        Jd[:, 0:3] = np.tile(np.eye(3), (nobs, 1))
        Jd[:, 3:3 + nf] = dm_pose[vis][:, :, free].reshape(3 * nobs, nf)
        if E:
            dm_shape = np.einsum('mab,mbe->mae', L, dv_shape.reshape(M, 9, E))
            Jd[:, 3 + nf:] = dm_shape[vis].reshape(3 * nobs, E)
        blocks.append(Jd * self.wt_data)
        if len(self.body_ids):
            _, _, Jp = gmm_prior_eval(self.prior, pose[self.body_ids], want_jac=True)
            Jb = np.zeros((Jp.shape[0], 3 + nf + E))
            pos = {int(pid): i for i, pid in enumerate(free)}
            for bi, pid in enumerate(self.body_ids):
                if int(pid) in pos:
                    Jb[:, 3 + pos[int(pid)]] = Jp[:, bi]
            blocks.append(Jb * self.wt_pose)
        if self.velo_target is not None:
            Jv = np.zeros((NP, 3 + nf + E))
            Jv[free, 3 + np.arange(len(free))] = 1.0
            blocks.append(Jv * self.wt_velo)
        if self.finger_ids is not None:
            Jh = np.zeros((len(self.finger_ids), 3 + nf + E))
            pos = {int(pid): i for i, pid in enumerate(free)}
            for hi, pid in enumerate(self.finger_ids):
                if int(pid) in pos:
                    Jh[hi, 3 + pos[int(pid)]] = 1.0
            blocks.append(Jh * self.wt_poseH)
        if self.face_ids is not None:
            Jf = np.zeros((len(self.face_ids), 3 + nf + E))
            pos = {int(pid): i for i, pid in enumerate(free)}
            for hi, pid in enumerate(self.face_ids):
                if int(pid) in pos:
                    Jf[hi, 3 + pos[int(pid)]] = 1.0
            blocks.append(Jf * self.wt_poseF)
        if self.shape_free:
            Je = np.zeros((E, 3 + nf + E))
            Je[np.arange(E), 3 + nf + np.arange(E)] = 1.0
            if self.shp_anchor is not None and self.wt_stay != 0.0:
                blocks.append(Je * self.wt_stay)
            blocks.append(Je * self.wt_shape)
        return np.vstack(blocks)


def stageii_chain(m, prior, closest, coef, obs, vis, model_type, weights=None, optimize_fingers=False,
                  optimize_toes=False, maxiter=100, reference_cost=False, init=None, num_train_markers=46,
                  collect_stats=False, optimize_face=False, free_shape=None):
    """The Stage-II frame loop, chmosh.py:584-724, on array inputs:
    obs[F,M,3] (metres), vis[F,M] bool (marker of latent label i observed in frame t).
    `init` = None -> first-frame schedule (rigid init + 3 annealed rounds, chmosh.py:629-655);
    or dict(pose, trans, pose_prev|None) to continue a chain (used for chunk tests).
    `optimize_face` adds the jaw pose ids + the `poseF` term to Step 2 (:685-686); `free_shape` = 'expr' | 'dmpl' frees
    the shape block declared with set_free_shape(m, ...) in Step 2: 'expr' -> regulariser stageii_wt_expr (:687-688);
    'dmpl' -> regulariser stageii_wt_dmpl plus `extrap_dmpl` (:693-699).  As written in the reference, `dmpl_prev` is
    refreshed (:658-659) BEFORE the term is built, so (dmpl - (dmpl.r + (dmpl.r - dmpl_prev))) * 6 evaluates to
    (dmpl - value at frame start) * 6, from the second solved frame on.
    Returns dict(fullpose[F',P], trans[F',3], markers_sim[list], frame_ids[F'], errs{term: array}, pose[F',NP],
    shape[F',E])."""
    W = stageii_weights_default() if weights is None else dict(weights)
    NP = m['NP']
    root, body, finger, step1, step2 = pose_id_sets(model_type, NP, optimize_fingers, optimize_toes, optimize_face)
    face = face_pose_ids(model_type, optimize_face)
    objf = StageIIObjective(m, closest, coef, prior, body, reference_cost=reference_cost)
    assert free_shape in (None, 'expr', 'dmpl')
    if free_shape is not None:
        assert 'E' in m, 'call set_free_shape(m, start, count) first'
    dmpl_prev_set = False
    M = closest.shape[0]
    F = obs.shape[0]
    pose_prev = None
    first = True
    if init is not None:
        objf.pose = np.array(init['pose'], dtype=np.float64)
        objf.trans = np.array(init['trans'], dtype=np.float64)
        pose_prev = None if init.get('pose_prev') is None else np.array(init['pose_prev'], dtype=np.float64)
        first = False
        if init.get('as_first_solved', False):
            # continue exactly as the reference would after its first solved frame: pose_prev stays None
            pass
    out = dict(fullpose=[], trans=[], markers_sim=[], frame_ids=[], pose=[], errs={}, iters=[])
    stats = {} if collect_stats else None
    for t in range(F):
        vmask = np.asarray(vis[t], dtype=bool)
        n_obs = int(vmask.sum())
        if n_obs == 0:
            continue  # chmosh.py:586-588
        n_miss = float(M - n_obs)
        anneal = 1.0
        if n_miss > 0:
            anneal = anneal + (n_miss / M) * W['stageii_wt_annealing']
        objf.vis = vmask
        objf.obs = np.asarray(obs[t], dtype=np.float64)
        objf.wt_data = W['stageii_wt_data'] * (num_train_markers / n_obs)
        wt_pose = W['stageii_wt_poseB'] * anneal
        objf.wt_pose = wt_pose
        objf.wt_poseH = W['stageii_wt_poseH'] * anneal
        objf.wt_velo = W['stageii_wt_velo']
        objf.finger_ids = None
        objf.face_ids = None
        objf.shape_free = False
        objf.shp_anchor = None
        objf.velo_target = None
        if pose_prev is not None:
            objf.velo_target = objf.pose + (objf.pose - pose_prev)  # chmosh.py:624-626
        st = {}
        if first:
            sim = objf.markers_sim()
            R, T = rigid_landmark_transform(sim[vmask].T, objf.obs[vmask].T)
            objf.pose[:3] = rotmat_to_rotvec(R)
            objf.trans[:] = T.ravel()
            for s in (10., 5., 1.):
                objf.wt_pose = s * wt_pose
                objf.free_ids = step1
                x = minimize_dogleg(objf, objf.x(), e_3=1e-3, delta_0=.5, maxiter=maxiter, stats=st)
                objf.set_x(x)
            first = False
        else:
            pose_prev = objf.pose.copy()
            dmpl_prev_set = True   # :658-659
        objf.wt_pose = wt_pose
        objf.free_ids = step1
        x = minimize_dogleg(objf, objf.x(), e_3=1e-2, delta_0=.5, maxiter=maxiter, stats=st)
        objf.set_x(x)
        if optimize_fingers:
            objf.finger_ids = np.asarray(finger, dtype=np.int64)
        if len(face):
            objf.face_ids = np.asarray(face, dtype=np.int64)
            objf.wt_poseF = W['stageii_wt_poseF'] * anneal
        if free_shape is not None:
            objf.shape_free = True
            objf.wt_shape = W['stageii_wt_expr'] if free_shape == 'expr' else W['stageii_wt_dmpl']
            if free_shape == 'dmpl' and dmpl_prev_set:
                objf.shp_anchor = objf.shp.copy()
                objf.wt_stay = 6.0
        objf.free_ids = step2
        x = minimize_dogleg(objf, objf.x(), e_3=1e-2, delta_0=.5, maxiter=maxiter, stats=st)
        objf.set_x(x)
        # record (chmosh.py:712-724)
        for k, v in objf.terms(objf.pose, objf.trans).items():
            out['errs'].setdefault(k, []).append(float(np.sum(v ** 2)))
        if objf.shp is not None:
            out.setdefault('shape', []).append(objf.shp.copy())
        out['markers_sim'].append(objf.markers_sim()[vmask].copy())
        out['fullpose'].append(fullpose_from_pose(m, objf.pose))
        out['trans'].append(objf.trans.copy())
        out['pose'].append(objf.pose.copy())
        out['frame_ids'].append(t)
        out['iters'].append(st.get('iterations', 0))
    res = dict(fullpose=np.array(out['fullpose']), trans=np.array(out['trans']), pose=np.array(out['pose']),
               markers_sim=out['markers_sim'], frame_ids=np.array(out['frame_ids'], dtype=np.int64),
               errs={k: np.array(v) for k, v in out['errs'].items()}, iters=np.array(out['iters']),
               shape=np.array(out['shape']) if 'shape' in out else None,
               final=dict(pose=objf.pose.copy(), trans=objf.trans.copy(),
                          pose_prev=None if pose_prev is None else pose_prev.copy()))
    return res
