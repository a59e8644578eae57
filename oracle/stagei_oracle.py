"""TEST INFRASTRUCTURE ONLY -- float64 NumPy restatement of MoSh++ Stage-I.

*** PARITY PARTLY PINNED *** (same situation as stageii_oracle.py: chumpy / psbody are absent, the reference has no tests).
Stage-I solves, jointly over a handful of picked frames (12 by default), for the subject's shape `betas`, the latent marker
positions on the canonical body and one pose + translation per frame (chmosh.py:83-455).  This file restates

  chmosh.py:57-80     prepare_mosh_markers_latent: markers on the vertex normal at `m2b_distance`, signed surface distance
  chmosh.py:180-224   TransformedCoeffs / TransformedLms wiring: the attachment (8-NN, local frame, coefficients) is re-evaluated at
                      every evaluation point and is differentiated through the canonical body (betas) and the latent markers
  chmosh.py:236-242   per-frame rigid initialisation
  chmosh.py:313-415   four annealing rounds of one dogleg each: data / poseB / init_* / beta / surf (+ poseH in the last two)
  chmosh.py:417-447   outputs: betas, markers_latent, nearest-vertex ids, per-term SSE
  scan2mesh/mesh_distance_main.py:187-297  signed point-to-mesh distance (`PtsToMesh(signed=True, normalize=False, rho=identity)`)
  scan2mesh/mesh_distance/sample2meshdist.h:67-205  closed-form distance derivatives per nearest part (plane / line / point)
  scan2mesh/ch_vert_normals.py:36-135, robustifiers.py:45-57  vertex / triangle normals, SignedSqrt

PINNED to reference code executed / compiled here (tests/test_ref_golden.py):
  * the closed-form distance gradients per nearest part  <- the reference's own C++ header sample2meshdist.h, compiled in place
    into oracle/_ref/libs2m_ref.so (oracle/ref_build/, with a stand-in for the few Eigen types it uses)
  * the normal chosen per part code, the sign rule, the signed square root  <- MeshDistanceSquared.direction and SignedSqrt method
    sources executed on our nearest-triangle data
  * (host package) marker_layout_load, the three frame pickers  <- the reference's functions
UNPINNED: the nearest-triangle search itself (psbody's AABB tree), chumpy's graph evaluation order / dogleg, psbody's SMPL.

Third-party pieces restated from their published behaviour: psbody.mesh `aabbtree_nearest` (closest point on a triangle mesh with
the part code 0 interior, 1-3 edges ab/bc/ca, 4-6 vertices a/b/c -- here an exhaustive search with Ericson's region tests) and
`Mesh.estimate_vertex_normals` (area-weighted sum of the incident triangle normals, normalised).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
"""
from __future__ import annotations

import numpy as np

from . import stageii_oracle as o2

__all__ = ['tri_normals_scaled', 'vert_normals', 'nearest_on_mesh', 'signed_surface_distance', 'markers_latent_init',
           'StageIObjective', 'stagei_solve', 'stagei_weights_default']


# ------------------------------------------------------------------------------------------
# normals (ch_vert_normals.py)
# ------------------------------------------------------------------------------------------
def tri_normals_scaled(v, f):
    """TriNormalsScaled (ch_vert_normals.py:79-80): (v1 - v0) x (v2 - v0) per face, length = 2 x area."""
    return np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])


def _normalize_rows(x):
    """NormalizedNx3 (ch_vert_normals.py:42-54): zero-length rows are divided by sqrt(1e-10)."""
    ss = (x * x).sum(1)
    ss = np.where(ss == 0, 1e-10, ss)
    return x / np.sqrt(ss)[:, None]


def vert_normals(v, f):
    """VertNormals(normalized=True) (ch_vert_normals.py:83-127) == psbody Mesh.estimate_vertex_normals: the sum of the scaled
    normals of the incident faces, normalised."""
    tn = tri_normals_scaled(v, f)
    vn = np.zeros_like(v)
    for c in range(3):
        np.add.at(vn, f[:, c], tn)
    return _normalize_rows(vn)


# ------------------------------------------------------------------------------------------
# closest point on a triangle mesh (psbody aabbtree_nearest restated; mesh_distance_main.py:330-352)
# ------------------------------------------------------------------------------------------
def _closest_on_triangles(p, a, b, c):
    """p[n,3] against triangles a,b,c[n,3] (row-wise).  Returns the closest point and the part code
    (0 interior, 1 ab, 2 bc, 3 ca, 4 a, 5 b, 6 c)."""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(1), (ac * ap).sum(1)
    bp = p - b
    d3, d4 = (ab * bp).sum(1), (ac * bp).sum(1)
    cp = p - c
    d5, d6 = (ab * cp).sum(1), (ac * cp).sum(1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    with np.errstate(divide='ignore', invalid='ignore'):
        t_ab = d1 / (d1 - d3)
        t_ca = d2 / (d2 - d6)
        t_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        den = 1.0 / (va + vb + vc)
    conds = [(d1 <= 0) & (d2 <= 0), (d3 >= 0) & (d4 <= d3), (vc <= 0) & (d1 >= 0) & (d3 <= 0), (d6 >= 0) & (d5 <= d6),
             (vb <= 0) & (d2 >= 0) & (d6 <= 0), (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)]
    parts = [4, 5, 1, 6, 3, 2]
    pts = [a, b, a + t_ab[:, None] * ab, c, a + t_ca[:, None] * ac, b + t_bc[:, None] * (c - b)]
    part = np.zeros(len(p), dtype=np.int64)
    q = a + ab * (vb * den)[:, None] + ac * (vc * den)[:, None]
    done = np.zeros(len(p), dtype=bool)
    for cnd, pc, pt in zip(conds, parts, pts):
        sel = cnd & ~done
        part[sel] = pc
        q[sel] = pt[sel]
        done |= sel
    return q, part


def nearest_on_mesh(pts, v, f):
    """For each point: nearest triangle id, part code, nearest point.  Exhaustive over the triangles that can hold the nearest
    point (all three vertices within d_nearest_vertex + longest edge of the point); ties go to the lowest face id."""
    pts = np.asarray(pts, dtype=np.float64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    lmax = np.sqrt(max(((a - b) ** 2).sum(1).max(), ((b - c) ** 2).sum(1).max(), ((c - a) ** 2).sum(1).max()))
    tri = np.zeros(len(pts), dtype=np.int64)
    part = np.zeros(len(pts), dtype=np.int64)
    near = np.zeros((len(pts), 3))
    for i, p in enumerate(pts):
        dv = np.sqrt(((v - p) ** 2).sum(1))
        lim = dv.min() + lmax
        cand = np.flatnonzero((dv[f] <= lim).any(1))
        q, pc = _closest_on_triangles(np.broadcast_to(p, (len(cand), 3)), a[cand], b[cand], c[cand])
        d2 = ((q - p) ** 2).sum(1)
        k = int(np.argmin(d2))
        tri[i], part[i], near[i] = cand[k], pc[k], q[k]
    return tri, part, near


def signed_surface_distance(pts, v, f, want_jac=False, vn=None):
    """PtsToMesh(signed=True, normalize=False, rho=identity) (mesh_distance_main.py:158-184): SignedSqrt(|p - nearest|^2 . direction),
    direction = sign((p - nearest) . n) with n the triangle normal (interior), the vertex normal (vertex) or the sum of the two
    vertex normals (edge) (:266-297).
    With want_jac also d/dp [M,3] and d/d(a,b,c) [M,3(vertex),3] of the nearest triangle's vertices -- sample2meshdist.h's pointPlane /
    pointLine / pointPoint derivatives composed with Square and SignedSqrt (the product 2 d . 1/(2 d) cancels), `direction` held
    constant as in MeshDistanceSquared.compute_dr_wrt (:250-252)."""
    pts = np.asarray(pts, dtype=np.float64)
    tri, part, near = nearest_on_mesh(pts, v, f)
    if vn is None:
        vn = vert_normals(v, f)
    fv = f[tri]                                     # M,3 vertex ids
    A, B, C = v[fv[:, 0]], v[fv[:, 1]], v[fv[:, 2]]
    nrm = np.cross(B - A, C - A)
    s = np.sqrt((nrm * nrm).sum(1))
    nh = nrm / s[:, None]
    diff = pts - near
    nn = np.zeros_like(pts)
    interior = part == 0
    vert = part > 3
    edge = (part > 0) & (part <= 3)
    nn[interior] = _normalize_rows(nrm[interior])
    nn[vert] = vn[fv[vert, part[vert] - 4]]
    nn[edge] = vn[fv[edge, part[edge] - 1]] + vn[fv[edge, np.mod(part[edge], 3)]]
    direction = np.sign((diff * nn).sum(1))
    d2 = (diff * diff).sum(1)
    dist = np.sqrt(d2) * direction
    if not want_jac:
        return dist, tri, part
    M = len(pts)
    dp = np.zeros((M, 3))
    dabc = np.zeros((M, 3, 3))
    # interior: h = (p - a) . nh ; dh/dp = nh ; through the normal: u = (I - nh nh^T)(p - a)/s
    if interior.any():
        i = np.flatnonzero(interior)
        pa = pts[i] - A[i]
        h = (pa * nh[i]).sum(1)
        u = (pa - h[:, None] * nh[i]) / s[i][:, None]
        e1, e2 = B[i] - A[i], C[i] - A[i]
        gb = np.cross(e2, u)
        gc = np.cross(u, e1)
        sg = (np.sign(h) * direction[i])[:, None]     # |h| then the sign: equals +1 whenever direction follows the face normal
        dp[i] = sg * nh[i]
        dabc[i, 1] = sg * gb
        dabc[i, 2] = sg * gc
        dabc[i, 0] = sg * (-nh[i] - gb - gc)
    with np.errstate(divide='ignore', invalid='ignore'):
        uh = diff / np.sqrt(d2)[:, None]
    uh = np.nan_to_num(uh)
    if edge.any():
        i = np.flatnonzero(edge)
        i0 = part[i] - 1
        i1 = np.mod(part[i], 3)
        P = v[fv[i, i0]]
        Q = v[fv[i, i1]]
        t = ((near[i] - P) * (Q - P)).sum(1) / ((Q - P) ** 2).sum(1)
        dp[i] = direction[i][:, None] * uh[i]
        dabc[i, i0] = -(direction[i] * (1 - t))[:, None] * uh[i]
        dabc[i, i1] = -(direction[i] * t)[:, None] * uh[i]
    if vert.any():
        i = np.flatnonzero(vert)
        dp[i] = direction[i][:, None] * uh[i]
        dabc[i, part[i] - 4] = -direction[i][:, None] * uh[i]
    return dist, tri, part, dp, dabc, fv


def markers_latent_init(can_v, faces, vids, m2b):
    """prepare_mosh_markers_latent (chmosh.py:57-67): vertex + vertex normal x distance-from-skin."""
    vn = vert_normals(can_v, faces)
    return can_v[vids] + vn[vids] * np.asarray(m2b, dtype=np.float64)[:, None]


# ------------------------------------------------------------------------------------------
# the Stage-I objective
# ------------------------------------------------------------------------------------------
def stagei_weights_default():
    """opt_weights.smplh / smplx of the reference configuration (moshpp_conf.yaml:103-125)."""
    return dict(stagei_wt_poseH=3.0, stagei_wt_poseF=3.0, stagei_wt_expr=34.0, stagei_wt_pose=3.0, stagei_wt_poseB=3.0,
                stagei_wt_init_finger_left=400.0, stagei_wt_init_finger_right=400.0, stagei_wt_init_finger=400.0,
                stagei_wt_betas=10.0, stagei_wt_init=300.0, stagei_wt_data=75.0, stagei_wt_surf=10000.0,
                stagei_wt_annealing=[1.0, 0.5, 0.25, 0.125])


def _frame_unit_jacs(v0, v1, v2):
    """df_i/d(v0,v1,v2) [M,3(i),3,9] of the three frame vectors, from markers_from_verts with unit coefficients."""
    out = []
    eye0 = np.zeros((3, 9)); eye0[:, :3] = np.eye(3)
    for i in range(3):
        c = np.zeros((len(v0), 3)); c[:, i] = 1.0
        _, L = o2.markers_from_verts(c, v0, v1, v2, want_jac=True)
        out.append(L - eye0[None])
    return np.stack(out, axis=1)


def _frames(v0, v1, v2):
    e1, e2 = v1 - v0, v2 - v0
    f1 = e1 / np.sqrt((e1 * e1).sum(1))[:, None]
    n = np.cross(e1, e2)
    f2 = n / np.sqrt((n * n).sum(1))[:, None]
    return np.stack([f1, f2, np.cross(f1, f2)], axis=1)      # M,3(i),3


class StageIObjective:
    """x = [trans_f (3 each), markers_latent (3M), pose_f[ids] (each frame), betas[:nb]] -- the order of `free_vars`
    (chmosh.py:383, 400-403).  `m` is a prepare_model()'d model with set_free_shape(m, 0, nb) (frozen betas zero in that block)."""

    def __init__(self, m, faces, prior, body_ids, frames, markers_latent, m2b, init_terms, nb, exclude_vids=None,
                 head_corr=None):
        self.m, self.faces, self.prior, self.body_ids = m, np.asarray(faces, dtype=np.int64), prior, list(body_ids)
        self.frames = [(np.asarray(ids, dtype=np.int64), np.asarray(obs, dtype=np.float64)) for ids, obs in frames]
        self.F = len(self.frames)
        self.ml = np.array(markers_latent, dtype=np.float64)
        self.M = self.ml.shape[0]
        self.m2b = np.asarray(m2b, dtype=np.float64)
        self.init_terms = [(np.asarray(ids, dtype=np.int64), float(w)) for ids, w in init_terms]   # wt before annealing
        self.nb = int(nb)
        self.exclude_vids = exclude_vids
        self.head_corr = head_corr            # (ids, C[h', h]) or None  (chmosh.py:252-266, 362-369)
        self.pose = np.zeros((self.F, m['NP']))
        self.trans = np.zeros((self.F, 3))
        self.betas = np.zeros(self.nb)
        self.pose_ids = []
        self.finger_ids = []
        self.face_ids = []
        self.w = {}
        # per-frame expression coefficients (optimize_face; chmosh.py:295-305, 394-398): the model's free shape block then holds the
        # expression columns and nb must be 0 (the reference cannot share betas and free expressions either, :295-299)
        self.E = 0
        self.expr = np.zeros((self.F, 0))
        self.expr_on = False
        self.fp_can = o2.fullpose_from_pose(m, np.zeros(m['NP']))
        # init markers: coefficients frozen at the start values, canonical body live (chmosh.py:188-190)
        can = self.can_verts(self.betas)
        self.cl0, self.coef0 = o2.transformed_coeffs(can, self.ml, exclude_vids)

    # -- pieces -------------------------------------------------------------------------------------------------------
    def can_verts(self, betas, vids=None):
        return o2.verts_forward(self.m, self.fp_can, np.zeros(3), vids, shp=betas if self.nb else None)

    def set_expressions(self, n_expr):
        assert self.nb == 0, 'free expressions need fixed betas'
        self.E = int(n_expr)
        self.expr = np.zeros((self.F, self.E))

    def set_round(self, pose_ids, finger_ids, w, face_ids=(), expr_on=False):
        """w: dict(data, poseB, poseH, beta, surf, anneal[, poseF, expr]) -- the per-round weights of chmosh.py:318-330."""
        self.pose_ids, self.finger_ids, self.w = list(pose_ids), list(finger_ids), dict(w)
        self.face_ids, self.expr_on = list(face_ids), bool(expr_on) and self.E > 0

    def x(self):
        parts = [self.trans.ravel(), self.ml.ravel(), self.pose[:, self.pose_ids].ravel(), self.betas]
        if self.expr_on:
            parts.append(self.expr.ravel())
        return np.concatenate(parts)

    def _unpack(self, x):
        F, M, npid = self.F, self.M, len(self.pose_ids)
        trans = x[:3 * F].reshape(F, 3)
        ml = x[3 * F:3 * F + 3 * M].reshape(M, 3)
        pose = self.pose.copy()
        pose[:, self.pose_ids] = x[3 * F + 3 * M:3 * F + 3 * M + F * npid].reshape(F, npid)
        o_b = 3 * F + 3 * M + F * npid
        betas = x[o_b:o_b + self.nb]
        self._expr_x = x[o_b + self.nb:].reshape(F, self.E) if self.expr_on else self.expr
        return trans, ml, pose, betas

    def set_x(self, x):
        self.trans, self.ml, self.pose, self.betas = [np.array(a) for a in self._unpack(x)]
        self.expr = np.array(self._expr_x)

    def markers_sim(self, f, ml=None, pose=None, trans=None, betas=None):
        ml = self.ml if ml is None else ml
        betas = self.betas if betas is None else betas
        can = self.can_verts(betas)
        cl, coef = o2.transformed_coeffs(can, ml, self.exclude_vids)
        fp = o2.fullpose_from_pose(self.m, self.pose[f] if pose is None else pose)
        shp = self.expr[f] if self.E else (betas if self.nb else None)
        v = o2.verts_forward(self.m, fp, self.trans[f] if trans is None else trans, cl.reshape(-1), shp=shp).reshape(-1, 3, 3)
        return o2.markers_from_verts(coef, v[:, 0], v[:, 1], v[:, 2])

    def evaluate(self, x, want_J=False):
        """Residual dict (term -> vector) and, optionally, the dense Jacobian dict (term -> [rows, n])."""
        m, F, M, nb = self.m, self.F, self.M, self.nb
        trans, ml, pose, betas = self._unpack(x)
        expr, E = self._expr_x, self.E
        npid = len(self.pose_ids)
        n = len(x)
        o_ml, o_pose, o_b = 3 * F, 3 * F + 3 * M, 3 * F + 3 * M + F * npid
        o_e = o_b + nb
        shp = betas if nb else None
        w = self.w
        can = self.can_verts(betas)
        cl, coef = o2.transformed_coeffs(can, ml, self.exclude_vids)          # re-evaluated at every point (TransformedCoeffs.on_changed)
        flat = cl.reshape(-1)
        vc = can[flat].reshape(M, 3, 3)
        res, jac = {}, {}
        if want_J:
            Fc = _frames(vc[:, 0], vc[:, 1], vc[:, 2])                         # canonical frame vectors M,3(i),3
            diffc = ml - vc[:, 0]
            if nb:
                _, _, dcan = o2.verts_jacobian(m, self.fp_can, np.zeros(3), flat, shp=shp, want_shape=True)
                dcan = dcan.reshape(M, 9, nb)                                   # d(v0c,v1c,v2c)/dbeta
                dF = _frame_unit_jacs(vc[:, 0], vc[:, 1], vc[:, 2])            # M,3(i),3,9
                # dc_i/dV = diff . df_i/dV - f_i . [I 0 0]
                dc_dV = np.einsum('ma,miab->mib', diffc, dF)
                dc_dV[:, :, :3] -= Fc
                dc_db = np.einsum('mib,mbe->mie', dc_dV, dcan)                  # M,3(i),nb
            pm = o2.pose_map_matrix(m)[:, self.pose_ids]                       # P x npid
        # ---- data (chmosh.py:199-213, 340): (obs - sim) * wt_data, frame by frame
        rd, Jd = [], []
        for f, (ids, obs) in enumerate(self.frames):
            fp = o2.fullpose_from_pose(m, pose[f])
            shp_f = expr[f] if E else shp                          # per-frame expressions ride in the model's free shape block
            if want_J:
                if nb or self.expr_on:
                    v, dv, dvs = o2.verts_jacobian(m, fp, trans[f], flat, shp=shp_f, want_shape=True)
                else:
                    v, dv = o2.verts_jacobian(m, fp, trans[f], flat, shp=shp_f)
            else:
                v = o2.verts_forward(m, fp, trans[f], flat, shp=shp_f)
            v3 = v.reshape(M, 3, 3)
            if want_J:
                sim, L = o2.markers_from_verts(coef, v3[:, 0], v3[:, 1], v3[:, 2], want_jac=True)
            else:
                sim = o2.markers_from_verts(coef, v3[:, 0], v3[:, 1], v3[:, 2])
            rd.append(((obs - sim[ids]) * w['data']).ravel())
            if want_J:
                Jf = np.zeros((len(ids), 3, n))
                Fp = _frames(v3[:, 0], v3[:, 1], v3[:, 2])                     # posed frame vectors
                dvp = dv.reshape(M, 9, -1).dot(pm)                             # M,9,npid
                Jf[:, :, o_pose + f * npid:o_pose + (f + 1) * npid] = np.einsum('mab,mbp->map', L[ids], dvp[ids])
                Jf[:, :, 3 * f:3 * f + 3] = np.eye(3)[None]                    # d sim / d trans (L sums to identity over the 3 verts)
                dsim_dml = np.einsum('mia,mib->mab', Fp, Fc)                   # sum_i f'_i f_i^T
                for k, i in enumerate(ids):
                    Jf[k, :, o_ml + 3 * i:o_ml + 3 * i + 3] = dsim_dml[i]
                if nb:
                    Jb = np.einsum('mab,mbe->mae', L, dvs.reshape(M, 9, nb)) + np.einsum('mia,mie->mae', Fp, dc_db)
                    Jf[:, :, o_b:o_b + nb] = Jb[ids]
                if self.expr_on:                                            # the canonical body does not see a frame's expression
                    Je = np.einsum('mab,mbe->mae', L, dvs.reshape(M, 9, E))
                    Jf[:, :, o_e + f * E:o_e + (f + 1) * E] = Je[ids]
                Jd.append(-w['data'] * Jf.reshape(-1, n))
        res['data'] = np.concatenate(rd)
        if want_J:
            jac['data'] = np.vstack(Jd)
        # ---- poseB (chmosh.py:342-345)
        if self.prior is not None and len(self.body_ids):
            rp, Jp = [], []
            cols = {pid: k for k, pid in enumerate(self.pose_ids)}
            for f in range(F):
                if want_J:
                    r, _, J0 = o2.gmm_prior_eval(self.prior, pose[f, self.body_ids], want_jac=True)
                    Jf = np.zeros((len(r), n))
                    for k, pid in enumerate(self.body_ids):
                        if pid in cols:
                            Jf[:, o_pose + f * npid + cols[pid]] = J0[:, k]
                    Jp.append(Jf * w['poseB'])
                else:
                    r, _ = o2.gmm_prior_eval(self.prior, pose[f, self.body_ids])
                rp.append(r * w['poseB'])
            res['poseB'] = np.concatenate(rp)
            if want_J:
                jac['poseB'] = np.vstack(Jp)
        # ---- init_* (chmosh.py:351-370): (markers_latent - init(betas)) per marker type
        vi = can[self.cl0.reshape(-1)].reshape(M, 3, 3)
        if want_J and nb:
            init, L0 = o2.markers_from_verts(self.coef0, vi[:, 0], vi[:, 1], vi[:, 2], want_jac=True)
            _, _, dcan0 = o2.verts_jacobian(m, self.fp_can, np.zeros(3), self.cl0.reshape(-1), shp=shp, want_shape=True)
            dinit_db = np.einsum('mab,mbe->mae', L0, dcan0.reshape(M, 9, nb))
        else:
            init = o2.markers_from_verts(self.coef0, vi[:, 0], vi[:, 1], vi[:, 2])
        loss = ml - init
        if want_J:
            Jl = np.zeros((M, 3, n))
            for i in range(M):
                Jl[i, :, o_ml + 3 * i:o_ml + 3 * i + 3] = np.eye(3)
            if nb:
                Jl[:, :, o_b:o_b + nb] = -dinit_db
        head_ids = set(self.head_corr[0].tolist()) if self.head_corr is not None else set()
        for t, (ids, wt) in enumerate(self.init_terms):
            keep = np.array([i for i in ids if i not in head_ids], dtype=np.int64)
            res[f'init_{t}'] = (loss[keep] * wt * w['anneal']).ravel()
            if want_J:
                jac[f'init_{t}'] = Jl[keep].reshape(-1, n) * wt * w['anneal']
        if self.head_corr is not None:
            hid, C = self.head_corr
            res['init_head_corr'] = (C.dot(loss[hid]) * w['init_head']).ravel()
            if want_J:
                jac['init_head_corr'] = np.einsum('gh,han->gan', C, Jl[hid]).reshape(-1, n) * w['init_head']
        # ---- beta (chmosh.py:372)
        if nb:
            res['beta'] = betas * w['beta']
            if want_J:
                Jb = np.zeros((nb, n)); Jb[:, o_b:o_b + nb] = np.eye(nb) * w['beta']
                jac['beta'] = Jb
        # ---- surf (chmosh.py:69-80, 373)
        if want_J:
            dist, tri, part, dp, dabc, fv = signed_surface_distance(ml, can, self.faces, want_jac=True)
            Js = np.zeros((M, n))
            for i in range(M):
                Js[i, o_ml + 3 * i:o_ml + 3 * i + 3] = dp[i]
            if nb:
                _, _, dtri = o2.verts_jacobian(m, self.fp_can, np.zeros(3), fv.reshape(-1), shp=shp, want_shape=True)
                Js[:, o_b:o_b + nb] = np.einsum('mva,mvae->me', dabc, dtri.reshape(M, 3, 3, nb))
            jac['surf'] = Js * w['surf']
        else:
            dist, tri, part = signed_surface_distance(ml, can, self.faces)
        res['surf'] = (dist - self.m2b) * w['surf']
        # ---- poseH (chmosh.py:391-393)
        if len(self.finger_ids):
            res['poseH'] = (pose[:, self.finger_ids] * w['poseH']).ravel()
            if want_J:
                cols = {pid: k for k, pid in enumerate(self.pose_ids)}
                Jh = np.zeros((F, len(self.finger_ids), n))
                for f in range(F):
                    for k, pid in enumerate(self.finger_ids):
                        Jh[f, k, o_pose + f * npid + cols[pid]] = w['poseH']
                jac['poseH'] = Jh.reshape(-1, n)
        # ---- poseF / expr (chmosh.py:394-398): jaw pose and the per-frame expression coefficients, last two rounds
        if len(self.face_ids):
            res['poseF'] = (pose[:, self.face_ids] * w['poseF']).ravel()
            if want_J:
                cols = {pid: k for k, pid in enumerate(self.pose_ids)}
                Jh = np.zeros((F, len(self.face_ids), n))
                for f in range(F):
                    for k, pid in enumerate(self.face_ids):
                        Jh[f, k, o_pose + f * npid + cols[pid]] = w['poseF']
                jac['poseF'] = Jh.reshape(-1, n)
        if self.expr_on:
            res['expr'] = (expr * w['expr']).ravel()
            if want_J:
                Je = np.zeros((F * E, n)); Je[:, o_e:] = np.eye(F * E) * w['expr']
                jac['expr'] = Je
        return (res, jac) if want_J else res

    def r(self, x):
        res = self.evaluate(x)
        return np.concatenate([res[k] for k in sorted(res)])

    def J(self, x):
        res, jac = self.evaluate(x, want_J=True)
        return np.vstack([jac[k] for k in sorted(res)])


class _RigidAdjustment:
    """`ch.minimize(fun=data_obj, x0=[p[:3] for p in poses] + trans, ...)` (chmosh.py:230-232): the unweighted marker residuals
    of all picked frames as a function of every frame's root orientation and translation; everything else stays where it is.
    x = [trans_f (3 each), root_f (3 each)] -- the order does not matter to the dogleg (all its norms are permutation invariant)."""

    def __init__(self, obj):
        self.obj = obj
        obj.set_round([0, 1, 2], [], dict(anneal=1.0, data=1.0, poseB=0.0, poseH=0.0, beta=0.0, surf=0.0, init_head=0.0,
                                          poseF=0.0, expr=0.0))
        self.full = obj.x()
        F, M = obj.F, obj.M
        self.cols = np.concatenate([np.arange(3 * F), 3 * F + 3 * M + np.arange(3 * F)])

    def x(self):
        return self.full[self.cols].copy()

    def _full(self, x):
        full = self.full.copy()
        full[self.cols] = x
        return full

    def r(self, x):
        return self.obj.evaluate(self._full(x))['data']

    def J(self, x):
        _, jac = self.obj.evaluate(self._full(x), want_J=True)
        return jac['data'][:, self.cols]


class _CentralDifferences:
    """The same residual with its Jacobian taken by central differences (h = 1e-6), as the executed-reference fixture's stand-in
    for ch.minimize does (tests/golden/make_ref_stageii_golden.py: minimize): for comparing trajectories like with like -- the
    objective is only piecewise smooth (re-evaluated attachment / nearest triangle), so a differenced Jacobian and the analytic one
    can part ways at a stopping decision."""

    def __init__(self, obj, h=1e-6):
        self.obj, self.h = obj, h

    def r(self, x):
        return self.obj.r(x)

    def J(self, x):
        cols = []
        for i in range(len(x)):
            xp = x.copy(); xp[i] += self.h
            xm = x.copy(); xm[i] -= self.h
            cols.append((self.obj.r(xp) - self.obj.r(xm)) / (2 * self.h))
        self.obj.r(x)
        return np.array(cols).T


def stagei_solve(m, faces, prior, model_type, frames, marker_vids, marker_type_mask, m2b_distance, nb, weights=None,
                 optimize_fingers=False, optimize_toes=False, betas_init=None, maxiter=100, stagei_lr=1e-3, exclude_vids=None,
                 head_corr=None, stats=None, optimize_face=False, expr_start=None, n_expr=0,
                 extra_initial_rigid_adjustment=False, difference_jacobian=False):
    """mosh_stagei's numeric core (chmosh.py:177-447).  `frames`: list of (latent marker ids, obs[n,3]) -- the `common_labels`
    selection of :199-206 already applied; `marker_vids`[M]; `marker_type_mask`: {type: bool[M]}; `m2b_distance`: {type: metres}.
    Returns betas, markers_latent, markers_latent_vids, per-frame pose / trans, per-term SSE of the last round."""
    W = stagei_weights_default() if weights is None else weights
    M = len(marker_vids)
    if optimize_face:
        assert nb == 0, 'optimize_face needs fixed betas (chmosh.py:295-299)'
        o2.set_free_shape(m, expr_start, n_expr)          # the free block now holds the expression columns, one value set per frame
    else:
        o2.set_free_shape(m, 0, nb)
    m2b = np.ones(M) * 0.0095
    for k, mask in marker_type_mask.items():
        m2b[np.asarray(mask, dtype=bool)] = m2b_distance[k]
    b0 = np.zeros(nb) if betas_init is None else np.asarray(betas_init, dtype=np.float64)[:nb]
    fp_can = o2.fullpose_from_pose(m, np.zeros(m['NP']))
    can = o2.verts_forward(m, fp_can, np.zeros(3), None, shp=b0 if nb else None)
    ml0 = markers_latent_init(can, np.asarray(faces), np.asarray(marker_vids), m2b)
    root, body, finger, _, _ = o2.pose_id_sets(model_type, m['NP'], optimize_fingers=optimize_fingers, optimize_toes=optimize_toes)
    init_terms = [(np.flatnonzero(np.asarray(mask, dtype=bool)), W.get(f'stagei_wt_init_{k}', W['stagei_wt_init']))
                  for k, mask in marker_type_mask.items()]
    obj = StageIObjective(m, faces, prior, body, frames, ml0, m2b, init_terms, nb, exclude_vids=exclude_vids, head_corr=head_corr)
    obj.betas = b0.copy()
    if optimize_face:
        obj.set_expressions(n_expr)
    face = o2.face_pose_ids(model_type, optimize_face)
    obj.cl0, obj.coef0 = o2.transformed_coeffs(obj.can_verts(obj.betas), obj.ml, exclude_vids)
    # rigid initialisation per frame (chmosh.py:236-238, rigid_transformations.py:73-83)
    for f, (ids, obs) in enumerate(obj.frames):
        sim = obj.markers_sim(f)[ids]
        R, T = o2.rigid_landmark_transform(sim.T, obs.T)
        obj.pose[f, :3] = o2.rotmat_to_rotvec(R)
        obj.trans[f] = np.asarray(T).ravel()
    if extra_initial_rigid_adjustment:   # chmosh.py:230-232
        adj = _RigidAdjustment(obj)
        xa = o2.minimize_dogleg(_CentralDifferences(adj) if difference_jacobian else adj, adj.x(), e_3=.001, delta_0=0.5, maxiter=maxiter, stats=stats)
        obj.set_x(adj._full(xa))
    anneal = list(W['stagei_wt_annealing'])
    res = None
    for tidx, a in enumerate(anneal):
        detailed = tidx > len(anneal) - 3
        w = dict(anneal=a, data=(W['stagei_wt_data'] / a) * (46.0 / M), poseB=W['stagei_wt_poseB'] * a,
                 poseH=W['stagei_wt_poseH'] * a, beta=W['stagei_wt_betas'] * a, surf=W['stagei_wt_surf'],
                 init_head=W.get('stagei_wt_init_body', W['stagei_wt_init']) * a, poseF=W['stagei_wt_poseF'] * a, expr=W['stagei_wt_expr'] * a)
        pose_ids = list(root) + list(body)
        if len(body) and not optimize_toes:
            pose_ids = sorted(set(pose_ids).difference(range(30, 36)))
        fing = list(finger) if (detailed and optimize_fingers) else []
        fc = list(face) if detailed else []
        pose_ids = sorted(set(pose_ids + fing + fc))
        obj.set_round(pose_ids, fing, w, face_ids=fc, expr_on=detailed and optimize_face)
        x = o2.minimize_dogleg(_CentralDifferences(obj) if difference_jacobian else obj, obj.x(), e_3=stagei_lr, delta_0=0.5, maxiter=maxiter, stats=stats)
        obj.set_x(x)
        res = obj.evaluate(x)
    can = obj.can_verts(obj.betas)
    d2 = ((obj.ml[:, None, :] - can[None]) ** 2).sum(-1)
    return dict(expression=obj.expr.copy(), betas=obj.betas.copy(), markers_latent=obj.ml.copy(), markers_latent_vids=np.argmin(d2, axis=1),
                pose=obj.pose.copy(), trans=obj.trans.copy(), errs={k: float((v ** 2).sum()) for k, v in res.items()},
                objective=obj)
