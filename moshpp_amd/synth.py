"""Seeded synthetic inputs for tests and bench.

No licensed SMPL-family model file, pose prior or c3d capture exists in this environment, so
every workload is generated: a procedural humanoid with the *exact* topology sizes of the named
model family (V, K, kinematic tree, posedirs width, hand-PCA layout), a max-mixture pose prior,
smooth ground-truth motion and marker observations (noise, dropouts, gaps).

This module is a data generator only: it carries its own small NumPy LBS so that it never
touches `oracle/` (product code must not) and never needs the GPU.
"""
from __future__ import annotations

import numpy as np

MODEL_DIMS = {  # V, K   (smpl_fast_derivatives.py:63-68 infers the type from posedirs.shape[2]//3)
    'smpl': (6890, 24),
    'smplh': (6890, 52),
    'smplx': (10475, 55),
    'mano': (778, 16),
}

_SMPL_BODY_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19]
_HAND_LOCAL = [-1, 0, 1, -1, 3, 4, -1, 6, 7, -1, 9, 10, -1, 12, 13]  # -1 = wrist


def kintree_parents(model_type):
    """Public kinematic trees of the SMPL family (parents[0] = -1)."""
    if model_type == 'smpl':
        return np.array(_SMPL_BODY_PARENTS + [20, 21], dtype=np.int64)
    if model_type == 'smplh':
        p = list(_SMPL_BODY_PARENTS)
        for wrist, base in ((20, 22), (21, 37)):
            p += [wrist if q < 0 else base + q for q in _HAND_LOCAL]
        return np.array(p, dtype=np.int64)
    if model_type == 'smplx':
        p = list(_SMPL_BODY_PARENTS) + [15, 15, 15]  # jaw, left eye, right eye
        for wrist, base in ((20, 25), (21, 40)):
            p += [wrist if q < 0 else base + q for q in _HAND_LOCAL]
        return np.array(p, dtype=np.int64)
    if model_type == 'mano':
        return np.array([-1] + [0 if q < 0 else 1 + q for q in _HAND_LOCAL], dtype=np.int64)
    raise ValueError(model_type)


_BODY_JOINTS = np.array([
    [0.00, 0.00, 0.00], [0.07, -0.09, 0.00], [-0.07, -0.09, 0.00], [0.00, 0.11, -0.02],
    [0.10, -0.47, 0.00], [-0.10, -0.47, 0.00], [0.00, 0.25, 0.00], [0.09, -0.87, -0.03],
    [-0.09, -0.87, -0.03], [0.00, 0.30, 0.02], [0.11, -0.93, 0.09], [-0.11, -0.93, 0.09],
    [0.00, 0.51, -0.02], [0.08, 0.42, 0.00], [-0.08, 0.42, 0.00], [0.00, 0.60, 0.03],
    [0.19, 0.45, -0.01], [-0.19, 0.45, -0.01], [0.45, 0.45, -0.03], [-0.45, 0.45, -0.03],
    [0.70, 0.45, -0.03], [-0.70, 0.45, -0.03]])


def _hand_joints(wrist, sign, scale=1.0):
    """15 finger joints fanning out from a wrist along sign*X."""
    out = []
    # index, middle, pinky, ring, thumb (MANO order); z offsets spread the fingers
    zoff = [0.030, 0.010, -0.030, -0.010, 0.045]
    base = [0.095, 0.100, 0.085, 0.095, 0.040]
    seg = [0.032, 0.034, 0.024, 0.030, 0.030]
    for f in range(5):
        for s in range(3):
            x = base[f] + seg[f] * s
            z = zoff[f] * (1.0 + 0.15 * s)
            y = -0.005 * s if f < 4 else -0.01 - 0.01 * s
            out.append(wrist + scale * np.array([sign * x, y, z]))
    return np.array(out)


def rest_joints(model_type):
    if model_type == 'smpl':
        return np.vstack([_BODY_JOINTS, [[0.78, 0.45, -0.03], [-0.78, 0.45, -0.03]]])
    if model_type == 'smplh':
        return np.vstack([_BODY_JOINTS, _hand_joints(_BODY_JOINTS[20], 1.0), _hand_joints(_BODY_JOINTS[21], -1.0)])
    if model_type == 'smplx':
        face = np.array([[0.0, 0.57, 0.07], [0.03, 0.64, 0.09], [-0.03, 0.64, 0.09]])
        return np.vstack([_BODY_JOINTS, face, _hand_joints(_BODY_JOINTS[20], 1.0),
                          _hand_joints(_BODY_JOINTS[21], -1.0)])
    if model_type == 'mano':
        return np.vstack([[[0.0, 0.0, 0.0]], _hand_joints(np.zeros(3), 1.0)])
    raise ValueError(model_type)


def _bone_radius(model_type, j, K):
    if model_type == 'mano':
        return 0.030 if j == 0 else 0.009
    body_r = {0: 0.13, 1: 0.08, 2: 0.08, 3: 0.13, 4: 0.06, 5: 0.06, 6: 0.14, 7: 0.045, 8: 0.045, 9: 0.14,
              10: 0.04, 11: 0.04, 12: 0.06, 13: 0.07, 14: 0.07, 15: 0.10, 16: 0.055, 17: 0.055,
              18: 0.045, 19: 0.045, 20: 0.035, 21: 0.035}
    if j in body_r:
        return body_r[j]
    if model_type == 'smpl':
        return 0.035
    if model_type == 'smplx' and j in (22, 23, 24):
        return 0.03 if j == 22 else 0.012
    return 0.008


def _segments(J, parents):
    """Per joint j: a segment starting at J_j pointing to the centroid of its children (leaf: short stub
    continuing the parent's direction)."""
    K = len(parents)
    kids = [[] for _ in range(K)]
    for j in range(1, K):
        kids[parents[j]].append(j)
    a = J.copy()
    b = np.zeros_like(J)
    for j in range(K):
        if kids[j]:
            b[j] = J[kids[j]].mean(0)
        else:
            d = J[j] - J[parents[j]] if parents[j] >= 0 else np.array([0.0, 0.05, 0.0])
            b[j] = J[j] + 0.6 * d
        if np.linalg.norm(b[j] - a[j]) < 1e-4:
            b[j] = a[j] + np.array([0.0, 0.02, 0.0])
    return a, b


def _dist_to_segments(pts, a, b):
    """pts[n,3], segments a[K,3]->b[K,3]: distances [n,K]."""
    ab = b - a
    l2 = (ab * ab).sum(1)
    ap = pts[:, None, :] - a[None]
    t = np.clip((ap * ab[None]).sum(-1) / l2[None], 0.0, 1.0)
    proj = a[None] + t[..., None] * ab[None]
    return np.sqrt(((pts[:, None, :] - proj) ** 2).sum(-1))


def _skin_and_blend(model_type, v_template, J0, seg_a, seg_b, rad, rng, num_betas, chunk):
    """Skinning weights, outward directions, joint regressor and blendshapes for a vertex set around the skeleton."""
    V, K = v_template.shape[0], J0.shape[0]
    # skinning weights: <= 4 nonzeros per vertex, rows sum to 1
    weights = np.zeros((V, K))
    outward = np.zeros((V, 3))
    for s in range(0, V, chunk):
        pts = v_template[s:s + chunk]
        d = _dist_to_segments(pts, seg_a, seg_b) / (rad[None] + 0.02)
        idx = np.argsort(d, axis=1)[:, :4]
        dd = np.take_along_axis(d, idx, axis=1)
        ww = np.exp(-(dd - dd[:, :1]) ** 2 / 0.08) * np.exp(-(dd - dd[:, :1]) / 0.25)
        ww[ww < 0.02] = 0.0
        ww /= ww.sum(1, keepdims=True)
        np.put_along_axis(weights[s:s + chunk], idx, ww, axis=1)
        # outward direction: away from nearest segment
        j0 = idx[:, 0]
        ab = seg_b[j0] - seg_a[j0]
        tpar = np.clip(((pts - seg_a[j0]) * ab).sum(1) / (ab * ab).sum(1), 0, 1)
        foot = seg_a[j0] + tpar[:, None] * ab
        o = pts - foot
        outward[s:s + chunk] = o / np.maximum(np.linalg.norm(o, axis=1, keepdims=True), 1e-9)

    # joint regressor: normalised Gaussian over the ~32 nearest vertices of each joint (rows sum to 1)
    J_regressor = np.zeros((K, V))
    for j in range(K):
        d2 = ((v_template - J0[j]) ** 2).sum(1)
        nn = np.argsort(d2)[:32]
        ww = np.exp(-d2[nn] / (2 * (rad[j] + 0.02) ** 2))
        J_regressor[j, nn] = ww / ww.sum()

    # shape blendshapes: smooth fields, decaying amplitude
    shapedirs = np.zeros((V, 3, num_betas))
    for k in range(num_betas):
        amp = 0.012 / (1.0 + 0.35 * k)
        A = rng.normal(0, 1, (3, 3)) * 0.5
        Wf = rng.normal(0, 4.0, (3, 3))
        ph = rng.uniform(0, 2 * np.pi, 3)
        shapedirs[:, :, k] = amp * (v_template.dot(A.T) + 0.4 * np.sin(v_template.dot(Wf.T) + ph))

    # pose blendshapes: dense, localised near the driving joint (magnitude ~ mm)
    nfeat = 9 * (K - 1)
    posedirs = np.zeros((V, 3, nfeat))
    for k in range(1, K):
        d2 = ((v_template - J0[k]) ** 2).sum(1)
        sig = 0.10 if model_type != 'mano' else 0.02
        g = 0.002 * np.exp(-d2 / (2 * sig * sig)) + 2e-5
        posedirs[:, :, 9 * (k - 1):9 * k] = g[:, None, None] * rng.normal(0, 1, (V, 3, 9))

    return weights, outward, J_regressor, shapedirs, posedirs


def synth_model(model_type, seed=0, num_betas=16, chunk=2048, vertex_order='shuffled'):
    """A synthetic body model with the exact topology sizes of `model_type`, in the layout of the
    reference's model pickles (keys as read by smpl_fast_derivatives.py:52-166).
    vertex_order: 'shuffled' (default; every fixture and bench number is on it) gives the vertices random ids -- the worst case for
    anything that hopes consecutive vertices share joints; 'mesh' keeps them bone by bone, along each bone, the way a registered
    artist mesh numbers them (tools/joint_census.py compares the two)."""
    V, K = MODEL_DIMS[model_type]
    rng = np.random.default_rng(seed + 7919 * (list(MODEL_DIMS).index(model_type) + 1))
    parents = kintree_parents(model_type)
    J0 = rest_joints(model_type)
    assert J0.shape[0] == K, (J0.shape, K)
    seg_a, seg_b = _segments(J0, parents)
    rad = np.array([_bone_radius(model_type, j, K) for j in range(K)])
    length = np.linalg.norm(seg_b - seg_a, axis=1)
    area = (length + 2 * rad) * rad
    counts = np.floor(area / area.sum() * V).astype(int)
    counts = np.maximum(counts, 6)
    while counts.sum() > V:
        counts[np.argmax(counts)] -= 1
    while counts.sum() < V:
        counts[rng.integers(K)] += 1
    verts = []
    for j in range(K):
        n = counts[j]
        t = rng.uniform(-0.15, 1.15, n)
        axis = seg_b[j] - seg_a[j]
        axis_n = axis / np.linalg.norm(axis)
        ref = np.array([0.0, 0.0, 1.0]) if abs(axis_n[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
        u = np.cross(axis_n, ref); u /= np.linalg.norm(u)
        w = np.cross(axis_n, u)
        ang = rng.uniform(0, 2 * np.pi, n)
        rr = rad[j] * (0.9 + 0.2 * rng.random(n))
        # round the caps
        cap = np.clip(np.minimum(t + 0.15, 1.15 - t) / 0.15, 0.0, 1.0)
        rr = rr * np.sqrt(0.15 + 0.85 * cap)
        p = seg_a[j][None] + t[:, None] * axis[None] + rr[:, None] * (np.cos(ang)[:, None] * u[None]
                                                                         + np.sin(ang)[:, None] * w[None])
        verts.append(p)
    if vertex_order == 'mesh':
        verts = [p[np.argsort((p - seg_a[j]).dot((seg_b[j] - seg_a[j]) / np.linalg.norm(seg_b[j] - seg_a[j])))] for j, p in enumerate(verts)]
    v_template = np.vstack(verts)
    perm = rng.permutation(V)          # (drawn in both cases: the random stream behind it stays what it was)
    if vertex_order != 'mesh':
        v_template = v_template[perm]

    weights, outward, J_regressor, shapedirs, posedirs = _skin_and_blend(model_type, v_template, J0, seg_a, seg_b, rad, rng,
                                                                         num_betas, chunk)
    kintree_table = np.vstack([np.where(parents < 0, 4294967295, parents), np.arange(K)]).astype(np.int64)
    dd = dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, weights=weights,
              J_regressor=J_regressor, kintree_table=kintree_table, bs_style='lbs', bs_type='lrotmin',
              f=np.zeros((0, 3), dtype=np.int64), model_type=model_type,
              _outward=outward, _rest_joints=J0)
    if model_type == 'mano':
        q, _ = np.linalg.qr(rng.normal(0, 1, (45, 45)))
        dd['hands_components'] = q
        dd['hands_mean'] = rng.normal(0, 0.08, 45)
    return dd


def _capsule_mesh(a, b, R, n_target):
    """Closed capsule around the segment a->b: two poles, rings of `ns` vertices over both caps and the cylinder.
    Returns vertices [n,3] and outward-wound triangles [t,3]."""
    axis = b - a
    L = np.linalg.norm(axis)
    ax = axis / L
    ref = np.array([0.0, 0.0, 1.0]) if abs(ax[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
    u = np.cross(ax, ref); u /= np.linalg.norm(u)
    w = np.cross(ax, u)
    ns = int(np.clip(round(np.sqrt(max(n_target, 8) * 2 * np.pi * R / (L + np.pi * R))), 6, 48))
    nr = max(5, int(round((n_target - 2) / ns)))
    nh = max(2, int(round(nr * (0.5 * np.pi * R) / (L + np.pi * R))))     # rings per cap
    nc = max(1, nr - 2 * nh)                                               # rings on the cylinder (incl. both rims)
    rings = []
    for i in range(nh):                                                    # cap at a: polar angle from the pole
        ph = 0.5 * np.pi * (i + 1) / (nh + (0 if nc > 1 else 1))
        rings.append((a - R * np.cos(ph) * ax, R * np.sin(ph)))
    for i in range(nc):
        t = (i + 0.5) / nc if nc > 1 else 0.5
        rings.append((a + t * axis, R))
    for i in range(nh - 1, -1, -1):
        ph = 0.5 * np.pi * (i + 1) / (nh + (0 if nc > 1 else 1))
        rings.append((b + R * np.cos(ph) * ax, R * np.sin(ph)))
    th = 2 * np.pi * np.arange(ns) / ns
    circ = np.cos(th)[:, None] * u[None] + np.sin(th)[:, None] * w[None]
    v = [a - R * ax] + [c[None] + r * circ for c, r in rings] + [b + R * ax]
    v = np.vstack([x.reshape(-1, 3) for x in v])
    f = []
    ring0 = lambda i: 1 + i * ns
    for k in range(ns):
        k1 = (k + 1) % ns
        f.append([0, ring0(0) + k1, ring0(0) + k])
        last = ring0(len(rings) - 1)
        f.append([len(v) - 1, last + k, last + k1])
        for i in range(len(rings) - 1):
            p, q = ring0(i), ring0(i + 1)
            f.append([p + k, p + k1, q + k1])
            f.append([p + k, q + k1, q + k])
    f = np.array(f, dtype=np.int64)
    # wind every triangle outward (normal pointing away from the axis)
    cen = v[f].mean(1)
    tpar = np.clip((cen - a).dot(ax), 0, L)
    out = cen - (a[None] + tpar[:, None] * ax[None])
    nrm = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    flip = (nrm * out).sum(1) < 0
    f[flip] = f[flip][:, [0, 2, 1]]
    return v, f


def synth_mesh_model(model_type, seed=0, num_betas=16, n_verts=None, chunk=2048):
    """Like synth_model but with a triangulated surface (`f`), as Stage-I needs one (vertex normals, point-to-surface
    distance): one closed capsule per bone, vertex count close to (not exactly) the family's V.  Same skeleton, radii,
    skinning / regressor / blendshape construction as synth_model."""
    V, K = MODEL_DIMS[model_type]
    V = V if n_verts is None else int(n_verts)
    rng = np.random.default_rng(seed + 104729 * (list(MODEL_DIMS).index(model_type) + 1))
    parents = kintree_parents(model_type)
    J0 = rest_joints(model_type)
    seg_a, seg_b = _segments(J0, parents)
    rad = np.array([_bone_radius(model_type, j, K) for j in range(K)])
    length = np.linalg.norm(seg_b - seg_a, axis=1)
    area = (length + 2 * rad) * rad
    counts = np.maximum(np.floor(area / area.sum() * V).astype(int), 32)
    vs, fs, off = [], [], 0
    for j in range(K):
        v, f = _capsule_mesh(seg_a[j], seg_b[j], rad[j], counts[j])
        vs.append(v); fs.append(f + off); off += len(v)
    v_template = np.vstack(vs)
    v_template += rng.normal(0, 0.0006, v_template.shape)   # breaks the exact collinearity of the ring / generator grid
    faces = np.vstack(fs)
    weights, outward, J_regressor, shapedirs, posedirs = _skin_and_blend(model_type, v_template, J0, seg_a, seg_b, rad, rng,
                                                                         num_betas, chunk)
    kintree_table = np.vstack([np.where(parents < 0, 4294967295, parents), np.arange(K)]).astype(np.int64)
    dd = dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, weights=weights,
              J_regressor=J_regressor, kintree_table=kintree_table, bs_style='lbs', bs_type='lrotmin',
              f=faces, model_type=model_type, _outward=outward, _rest_joints=J0)
    # vertices not buried inside another bone's capsule: where markers may sit
    d = _dist_to_segments(v_template, seg_a, seg_b) - rad[None]
    own = np.repeat(np.arange(K), [len(x) for x in vs])
    d[np.arange(len(own)), own] = np.inf
    dd['_exposed'] = d.min(1) > 0.02
    if model_type == 'mano':
        q, _ = np.linalg.qr(rng.normal(0, 1, (45, 45)))
        dd['hands_components'] = q
        dd['hands_mean'] = rng.normal(0, 0.08, 45)
    return dd


def synth_hand_prior(seed=0):
    """Contents of `pose_hand_prior.npz` (smpl_fast_derivatives.py:84-93): per-hand 45x45 PCA + means."""
    rng = np.random.default_rng(seed + 4242)
    ql, _ = np.linalg.qr(rng.normal(0, 1, (45, 45)))
    qr, _ = np.linalg.qr(rng.normal(0, 1, (45, 45)))
    return dict(componentsl=ql, componentsr=qr, hands_meanl=rng.normal(0, 0.08, 45),
                hands_meanr=rng.normal(0, 0.08, 45))


def synth_gmm_prior(seed=0, n_gaussians=8, npose_full=69):
    """Contents of `pose_body_prior.pkl` (gmm_prior_ch.py:110-120): means G x 69, covars G x 69 x 69, weights G."""
    rng = np.random.default_rng(seed + 1717)
    means = rng.normal(0, 0.15, (n_gaussians, npose_full))
    covars = np.zeros((n_gaussians, npose_full, npose_full))
    for g in range(n_gaussians):
        q, _ = np.linalg.qr(rng.normal(0, 1, (npose_full, npose_full)))
        ev = np.exp(rng.uniform(np.log(0.01), np.log(0.3), npose_full))
        covars[g] = (q * ev).dot(q.T)
        covars[g] = 0.5 * (covars[g] + covars[g].T)
    w = rng.uniform(0.5, 1.5, n_gaussians)
    return dict(means=means, covars=covars, weights=w / w.sum())


# ------------------------------------------------------------------------------------------
# tiny NumPy LBS used only to synthesise observations
# ------------------------------------------------------------------------------------------
def _rodrigues(r):
    t = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0.0]])
    if t < 1e-8:
        return np.eye(3) + K
    return np.eye(3) + np.sin(t) / t * K + (1 - np.cos(t)) / (t * t) * K.dot(K)


def lbs_numpy(v_shaped, J, posedirs, weights, parents, fullpose, trans, vids=None):
    """v = sum_j w_j (Rw_j (v_posed - J_j) + tw_j) + trans for one frame."""
    K = len(parents)
    R = np.array([_rodrigues(fullpose[3 * j:3 * j + 3]) for j in range(K)])
    feat = (R[1:] - np.eye(3)).reshape(-1)
    sl = slice(None) if vids is None else vids
    vp = v_shaped[sl] + posedirs[sl].dot(feat)
    Rw = np.zeros((K, 3, 3)); tw = np.zeros((K, 3))
    Rw[0] = R[0]; tw[0] = J[0]
    for j in range(1, K):
        p = parents[j]
        Rw[j] = Rw[p].dot(R[j]); tw[j] = Rw[p].dot(J[j] - J[p]) + tw[p]
    off = tw - np.einsum('kab,kb->ka', Rw, J)
    w = weights[sl]
    T = np.einsum('nk,kab->nab', w, Rw)
    return np.einsum('nab,nb->na', T, vp) + w.dot(off) + trans


def attach_markers(can_body, markers_latent, exclude_vids=None):
    """8-NN local-frame attachment, same construction as transformed_lm.py:59-113 (generator copy)."""
    V = can_body.shape[0]
    keep = np.arange(V)
    if exclude_vids is not None and len(exclude_vids):
        mask = np.ones(V, bool); mask[np.asarray(exclude_vids)] = False
        keep = keep[mask]
    pts = can_body[keep]
    d2 = ((markers_latent[:, None] - pts[None]) ** 2).sum(-1)
    cl = np.argsort(d2, axis=1, kind='stable')[:, :3]
    v0, v1, v2 = pts[cl[:, 0]], pts[cl[:, 1]], pts[cl[:, 2]]
    f1 = (v1 - v0) / np.linalg.norm(v1 - v0, axis=1, keepdims=True)
    n = np.cross(v1 - v0, v2 - v0)
    f2 = n / np.linalg.norm(n, axis=1, keepdims=True)
    f3 = np.cross(f1, f2)
    d = markers_latent - v0
    coef = np.stack([(d * f1).sum(1), (d * f2).sum(1), (d * f3).sum(1)], 1)
    return keep[cl], coef


def markers_numpy(coef, v0, v1, v2):
    f1 = (v1 - v0) / np.linalg.norm(v1 - v0, axis=1, keepdims=True)
    n = np.cross(v1 - v0, v2 - v0)
    f2 = n / np.linalg.norm(n, axis=1, keepdims=True)
    f3 = np.cross(f1, f2)
    return v0 + coef[:, :1] * f1 + coef[:, 1:2] * f2 + coef[:, 2:3] * f3


def pick_marker_vids(dd, n_markers, seed=0, body_only=True):
    """Farthest-point sampling of marker vertices.  body_only: restrict to vertices whose dominant joint
    is a body joint (0..21) -- the 41/53-marker body layouts of BASELINE configs 1-2."""
    rng = np.random.default_rng(seed + 99)
    v = dd['v_template']
    dom = np.argmax(dd['weights'], axis=1)
    cand = np.arange(v.shape[0])
    if body_only and dd['model_type'] != 'mano':
        cand = cand[dom[cand] <= 21]
    first = cand[rng.integers(len(cand))]
    chosen = [first]
    dmin = ((v[cand] - v[first]) ** 2).sum(1)
    for _ in range(n_markers - 1):
        nxt = cand[int(np.argmax(dmin))]
        chosen.append(nxt)
        dmin = np.minimum(dmin, ((v[cand] - v[nxt]) ** 2).sum(1))
    return np.array(chosen, dtype=np.int64)


def synth_motion(NP, body_dof, n_frames, seed=0, fps=120.0, amp_body=0.45, amp_hand=0.25, ramp=60):
    """Smooth ground-truth pose[F,NP] (sinusoid mixtures) and root translation[F,3]."""
    rng = np.random.default_rng(seed + 31337)
    t = np.arange(n_frames) / fps
    pose = np.zeros((n_frames, NP))
    for d in range(NP):
        amp = (amp_body if d < body_dof else amp_hand) * rng.uniform(0.2, 1.0)
        if d < 3:
            amp *= 0.8
        acc = np.zeros(n_frames)
        for _ in range(3):
            f = rng.uniform(0.1, 1.2)
            acc += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
        pose[:, d] = amp * acc / 2.0
    env = np.clip(np.arange(n_frames) / float(max(ramp, 1)), 0.15, 1.0)[:, None]
    pose *= env
    pose[:, 30:36] = 0.0 if NP > 36 else pose[:, 30:36]  # feet flat: toes are not optimised by default
    trans = np.stack([0.6 * np.sin(2 * np.pi * 0.11 * t + 0.3), 0.9 + 0.05 * np.sin(2 * np.pi * 0.9 * t),
                      0.8 * np.sin(2 * np.pi * 0.07 * t + 1.1)], axis=1)
    return pose, trans


def make_sequence(model_type='smplh', n_frames=120, n_markers=53, seed=0, noise=0.0005, dropout=0.02,
                  n_gaps=2, num_betas=16, dof_per_hand=24, use_hands_mean=True, body_only_markers=True,
                  empty_frames=(), dd=None, motion_seed=None):
    """Everything one Stage-II call needs, generated from seeds (`motion_seed`: another capture -- motion, noise, dropouts -- of
    the SAME subject, i.e. the same model, betas, priors and marker placement as `seed` gives):
    model pickle dict, hand prior, GMM prior, betas, latent markers + labels, marker_meta,
    mocap markers[F,N,3] in metres (NaN where dropped) + labels, and the ground truth."""
    rng = np.random.default_rng(seed + 5)
    if dd is None:
        dd = synth_model(model_type, seed=seed, num_betas=num_betas)
    V, K = MODEL_DIMS[model_type]
    parents = kintree_parents(model_type)
    hand_prior = synth_hand_prior(seed) if model_type in ('smplh', 'smplx') else None
    gmm = synth_gmm_prior(seed)
    betas = rng.normal(0, 0.7, num_betas)
    v_shaped = dd['v_template'] + dd['shapedirs'].dot(betas)
    J = dd['J_regressor'].dot(v_shaped)
    # pose variable layout (smpl_fast_derivatives.py:80-128)
    if model_type in ('smplh', 'smplx'):
        body_dof = 3 * K - 90
        comps = np.zeros((2 * dof_per_hand, 90))
        comps[:dof_per_hand, :45] = hand_prior['componentsl'][:dof_per_hand]
        comps[dof_per_hand:, 45:] = hand_prior['componentsr'][:dof_per_hand]
        hmean = np.concatenate([hand_prior['hands_meanl'], hand_prior['hands_meanr']]) if use_hands_mean \
            else np.zeros(90)
        NP = body_dof + 2 * dof_per_hand
    elif model_type == 'mano':
        body_dof = 3
        comps = dd['hands_components'][:dof_per_hand]
        hmean = np.zeros(45) if use_hands_mean else dd['hands_mean']  # inverted on purpose (:114)
        NP = 3 + dof_per_hand
    else:
        body_dof = 3 * K
        comps = np.zeros((0, 0)); hmean = np.zeros(0)
        NP = body_dof

    def fullpose_of(pose):
        if NP == body_dof:
            return pose.copy()
        return np.concatenate([pose[:body_dof], hmean + pose[body_dof:].dot(comps)])

    can_body = lbs_numpy(v_shaped, J, dd['posedirs'], dd['weights'], parents, fullpose_of(np.zeros(NP)), np.zeros(3))
    vids = pick_marker_vids(dd, n_markers, seed=seed, body_only=body_only_markers)
    markers_latent = can_body[vids] + dd['_outward'][vids] * 0.0095
    labels = [f'MK{idx:02d}' for idx in range(n_markers)]
    closest, coef = attach_markers(can_body, markers_latent)
    pose_gt, trans_gt = synth_motion(NP, body_dof, n_frames, seed=seed if motion_seed is None else motion_seed)
    if motion_seed is not None:
        rng = np.random.default_rng(motion_seed + 5)
    if not (model_type in ('smplh', 'smplx') and not body_only_markers) and model_type != 'mano':
        pose_gt[:, body_dof:] = 0.0  # body-only layouts cannot observe the fingers
    if model_type == 'smplx':
        pose_gt[:, 66:75] = 0.0  # jaw / eyes are never free in round-1 scope
    markers = np.zeros((n_frames, n_markers, 3))
    flat = closest.reshape(-1)
    for f in range(n_frames):
        v = lbs_numpy(v_shaped, J, dd['posedirs'], dd['weights'], parents, fullpose_of(pose_gt[f]), trans_gt[f],
                      vids=flat).reshape(n_markers, 3, 3)
        markers[f] = markers_numpy(coef, v[:, 0], v[:, 1], v[:, 2])
    markers += rng.normal(0, noise, markers.shape)
    drop = rng.random((n_frames, n_markers)) < dropout
    for _ in range(n_gaps):
        mk = rng.integers(n_markers)
        s = rng.integers(max(1, n_frames - 10))
        drop[s:s + rng.integers(10, 50), mk] = True
    drop[0, :] = False  # a clean first frame, like a calibration pose
    for ef in empty_frames:
        drop[ef, :] = True
    markers[drop] = np.nan
    marker_meta = dict(marker_vids={l: int(v) for l, v in zip(labels, vids)},
                       marker_type={l: 'body' for l in labels},
                       marker_type_mask={'body': np.ones(n_markers, dtype=bool)},
                       m2b_distance={'body': 0.0095}, surface_model_type=model_type)
    return dict(model=dd, hand_prior=hand_prior, gmm=gmm, betas=betas, markers_latent=markers_latent,
                latent_labels=labels, marker_meta=marker_meta, markers=markers, labels=list(labels),
                frame_rate=120.0, pose_gt=pose_gt, trans_gt=trans_gt, model_type=model_type,
                dof_per_hand=dof_per_hand, use_hands_mean=use_hands_mean, num_betas=num_betas)


# ------------------------------------------------------------------------------------------
# Stage-I problems (generator only; carries its own NumPy pieces like the Stage-II generator above)
# ------------------------------------------------------------------------------------------
def vertex_normals(v, f):
    """Area-weighted vertex normals of a triangle mesh."""
    tn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    vn = np.zeros_like(v)
    for c in range(3):
        np.add.at(vn, f[:, c], tn)
    n = np.linalg.norm(vn, axis=1, keepdims=True)
    return vn / np.where(n == 0, 1.0, n)


def make_stagei_problem(model_type='smplh', n_verts=2500, nb=6, M=30, F=6, seed=0, dof_per_hand=12, finger_markers=False):
    """A seeded Stage-I problem on the triangulated synthetic body: model arrays with the pose-variable layout of the family,
    raw GMM prior, marker layout vertex ids on exposed body vertices, a ground-truth subject (betas, latent markers a few
    millimetres off their layout positions) and F observed frames `(latent ids, xyz[n,3])` with dropouts and noise."""
    rng = np.random.default_rng(seed)
    dd = synth_mesh_model(model_type, seed=seed, num_betas=10, n_verts=n_verts)
    K = dd['weights'].shape[1]
    parents = kintree_parents(model_type)
    if model_type in ('smplh', 'smplx'):
        hp = synth_hand_prior(seed)
        body_dof = 3 * K - 90
        comps = np.zeros((2 * dof_per_hand, 90))
        comps[:dof_per_hand, :45] = hp['componentsl'][:dof_per_hand]
        comps[dof_per_hand:, 45:] = hp['componentsr'][:dof_per_hand]
        hand_dof, hands_mean = 2 * dof_per_hand, np.zeros(90)
    elif model_type == 'mano':
        body_dof, hand_dof, hands_mean, comps = 3, dof_per_hand, dd['hands_mean'], dd['hands_components'][:dof_per_hand]
    else:
        body_dof, hand_dof, hands_mean, comps = 3 * K, 0, None, None
    NP = body_dof + hand_dof
    model = dict(v_template=dd['v_template'], shapedirs=dd['shapedirs'], posedirs=dd['posedirs'], weights=dd['weights'],
                 J_regressor=dd['J_regressor'], parents=parents, body_dof=body_dof, hand_dof=hand_dof,
                 hands_mean=hands_mean, selected_components=comps)

    def fullpose_of(pose):
        if hand_dof == 0:
            return pose[:body_dof].copy()
        return np.concatenate([pose[:body_dof], hands_mean + pose[body_dof:].dot(comps)])

    dom = np.argmax(dd['weights'], 1)
    ok = dd['_exposed'] & ((dom <= 21) | finger_markers) if model_type != 'mano' else np.ones(len(dom), bool)
    cand = np.flatnonzero(ok)
    v = dd['v_template']
    vids = [cand[rng.integers(len(cand))]]
    dmin = ((v[cand] - v[vids[0]]) ** 2).sum(1)
    for _ in range(M - 1):
        nxt = cand[int(np.argmax(dmin))]
        vids.append(nxt)
        dmin = np.minimum(dmin, ((v[cand] - v[nxt]) ** 2).sum(1))
    vids = np.array(vids)
    betas_gt = rng.normal(0, 0.8, nb)
    v_shaped = dd['v_template'] + dd['shapedirs'][:, :, :nb].dot(betas_gt)
    J = dd['J_regressor'].dot(v_shaped)
    can_gt = lbs_numpy(v_shaped, J, dd['posedirs'], dd['weights'], parents, fullpose_of(np.zeros(NP)), np.zeros(3))
    skin = 0.0095 if model_type != 'mano' else 0.003
    ml_gt = can_gt[vids] + vertex_normals(can_gt, dd['f'])[vids] * skin \
        + rng.normal(0, 0.004 if model_type != 'mano' else 0.001, (M, 3))
    cl, coef = attach_markers(can_gt, ml_gt)
    pose_gt, trans_gt = synth_motion(NP, body_dof, 400, seed=seed)
    frames = []
    for t in np.linspace(60, 399, F).astype(int):
        p = pose_gt[t].copy()
        if not finger_markers and model_type != 'mano':
            p[body_dof:] = 0
        if model_type != 'mano':
            p[30:36] = 0
        if model_type == 'smplx':
            p[66:75] = 0
        vv = lbs_numpy(v_shaped, J, dd['posedirs'], dd['weights'], parents, fullpose_of(p), trans_gt[t],
                       vids=cl.reshape(-1)).reshape(M, 3, 3)
        sim = markers_numpy(coef, vv[:, 0], vv[:, 1], vv[:, 2]) + rng.normal(0, 0.0003, (M, 3))
        ids = np.flatnonzero(rng.random(M) > 0.05)
        frames.append((ids, sim[ids]))
    return dict(model=model, faces=dd['f'], gmm=synth_gmm_prior(seed), frames=frames, vids=vids, betas_gt=betas_gt, ml_gt=ml_gt,
                nb=nb, M=M, skin=skin, model_type=model_type, dd=dd, NP=NP,
                attach_gt=(cl, coef), v_shaped_gt=v_shaped, J_gt=J, fullpose_of=fullpose_of)
