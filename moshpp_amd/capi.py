"""ctypes binding of libmoshii.so (include/moshii.h).

The library is the product: there is NO CPU fallback.  `load()` raises if the shared object is
missing, and every compute entry point returns MOSHII_ERR_NO_DEVICE without a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MOSHII_LIB', os.path.join(_HERE, 'libmoshii.so'))
# limits of the chain kernel (include/moshii.h "Limits"): one chain's whole state lives in the 160 KiB LDS of a CU
MAX_MARKERS = 128            # moshii_attach_create
MAX_JOINTS = 64              # moshii_model_create (ancestor sets are 64-bit masks)
MAX_UNKNOWNS = 127           # 3 + free pose variables per solve, plain kernel (8 register blocks of 16, one row for the right-hand side)
MAX_UNKNOWNS_EXTENDED = 207  # ... with the jaw term / a free shape block (13 register blocks)

BUFFERS_HOST = 0
BUFFERS_DEVICE = 1

_c_double_p = C.POINTER(C.c_double)
_c_float_p = C.POINTER(C.c_float)
_c_int_p = C.POINTER(C.c_int32)
_c_u8_p = C.POINTER(C.c_uint8)


class ModelDesc(C.Structure):
    _fields_ = [('V', C.c_int32), ('K', C.c_int32), ('NB', C.c_int32), ('body_dof', C.c_int32),
                ('hand_dof', C.c_int32), ('parents', _c_int_p), ('v_template', _c_double_p),
                ('shapedirs', _c_double_p), ('posedirs', _c_double_p), ('weights', _c_double_p),
                ('J_regressor', _c_double_p), ('hands_mean', _c_double_p), ('selected_components', _c_double_p)]


class SolveOpts(C.Structure):
    _fields_ = [('wt_data', C.c_double), ('wt_velo', C.c_double), ('wt_poseB', C.c_double),
                ('wt_poseH', C.c_double), ('wt_annealing', C.c_double), ('num_train_markers', C.c_double),
                ('e3_first', C.c_double), ('e3', C.c_double), ('delta0', C.c_double), ('maxiter', C.c_int32),
                ('n_step1', C.c_int32), ('step1_ids', _c_int_p), ('n_step2', C.c_int32), ('step2_ids', _c_int_p),
                ('n_body', C.c_int32), ('body_ids', _c_int_p), ('n_finger', C.c_int32), ('finger_ids', _c_int_p),
                ('n_face', C.c_int32), ('face_ids', _c_int_p), ('wt_poseF', C.c_double), ('n_shape', C.c_int32),
                ('wt_shape', C.c_double), ('wt_shape_stay', C.c_double)]


NERR = 8   # MOSHII_NERR: data, poseB, velo, poseH, poseF, shape, shape_stay, 0


class ChainDesc(C.Structure):
    _fields_ = [('attach', C.c_void_p), ('F', C.c_int32), ('first_frame_schedule', C.c_int32),
                ('obs', C.c_void_p), ('vis', C.c_void_p), ('init_pose', _c_double_p), ('init_trans', _c_double_p),
                ('init_pose_prev', _c_double_p), ('init_shape', _c_double_p), ('pose', C.c_void_p),
                ('fullpose', C.c_void_p), ('trans', C.c_void_p), ('markers_sim', C.c_void_p), ('errs', C.c_void_p),
                ('iters', C.c_void_p), ('status', C.c_void_p), ('shape', C.c_void_p)]


class SequenceDesc(C.Structure):
    _fields_ = [('attach', C.c_void_p), ('F', C.c_int32), ('obs', C.c_void_p), ('vis', C.c_void_p),
                ('init_pose', _c_double_p), ('init_trans', _c_double_p), ('init_pose_prev', _c_double_p),
                ('pose', C.c_void_p), ('fullpose', C.c_void_p), ('trans', C.c_void_p), ('markers_sim', C.c_void_p),
                ('errs', C.c_void_p), ('iters', C.c_void_p), ('status', C.c_void_p),
                ('init_shape', _c_double_p), ('shape', C.c_void_p)]


class ChunkOpts(C.Structure):
    _fields_ = [('num_chunks', C.c_int32), ('warmup', C.c_int32), ('verify_tol', C.c_double)]


class ChunkReport(C.Structure):
    _fields_ = [('n_chunks', C.c_int32), ('n_repaired', C.c_int32), ('repair_rounds', C.c_int32), ('warmup', C.c_int32),
                ('max_handoff_dev', C.c_double), ('verify_tol', C.c_double)]


EXPORTS = {
    # name: (restype, argtypes)
    'moshii_last_error': (C.c_char_p, []),
    'moshii_version': (C.c_int, []),
    'moshii_source_hash': (C.c_char_p, []),
    'moshii_device_count': (C.c_int, []),
    'moshii_set_device': (C.c_int, [C.c_int]),
    'moshii_device_multiprocessors': (C.c_int, []),
    'moshii_model_create': (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    'moshii_model_destroy': (C.c_int, [C.c_void_p]),
    'moshii_model_set_betas': (C.c_int, [C.c_void_p, _c_double_p, C.c_int32]),
    'moshii_model_set_free_shape': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    'moshii_model_get_joints': (C.c_int, [C.c_void_p, _c_double_p]),
    'moshii_lbs_forward_f64': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    'moshii_lbs_forward_f32': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    'moshii_prior_create': (C.c_int, [C.c_int32, C.c_int32, _c_double_p, _c_double_p, _c_double_p, C.POINTER(C.c_void_p)]),
    'moshii_prior_destroy': (C.c_int, [C.c_void_p]),
    'moshii_attach_create': (C.c_int, [C.c_void_p, C.c_int32, _c_int_p, _c_double_p, C.POINTER(C.c_void_p)]),
    'moshii_attach_destroy': (C.c_int, [C.c_void_p]),
    'moshii_attach_markers': (C.c_int, [C.c_void_p, C.c_int32, _c_double_p, _c_double_p, _c_double_p]),
    'moshii_chain_solve': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(SolveOpts), C.c_int32, C.POINTER(ChainDesc),
                                     C.c_uint32, C.c_void_p]),
    'moshii_plan_chunks': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _c_int_p, _c_int_p]),
    'moshii_sequence_solve': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(SolveOpts), C.c_int32, C.POINTER(SequenceDesc),
                                        C.POINTER(ChunkOpts), C.c_uint32, C.c_void_p, C.POINTER(ChunkReport)]),
    'moshii_last_launch_info': (C.c_int, [C.c_char_p, C.c_int32, _c_int_p, _c_int_p]),
}

_lib = None


class MoshiiError(RuntimeError):
    pass


def load():
    """dlopen libmoshii.so (built in-tree by __graft_entry__.build() / moshpp_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MoshiiError(f'{LIB_PATH} not found: build it with `python -m moshpp_amd.build` '
                          f'(there is no CPU fallback for the Stage-II path)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().moshii_last_error()
        raise MoshiiError(f'libmoshii error {rc}: {msg.decode() if msg else "?"}')


def device_count():
    return load().moshii_device_count()


def device_cu_count():
    """CUs of the current device (256 on MI355X); the default chunk count of the chunked chain modes."""
    return max(int(load().moshii_device_multiprocessors()), 1)


def require_device():
    if device_count() < 1:
        raise MoshiiError('no HIP device visible: the moshpp_amd Stage-II path runs only on the GPU')


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(_c_double_p)


def _ip(a):
    return a.ctypes.data_as(_c_int_p)


class Model:
    """moshii_model_t: body model arrays + pose-variable layout, resident in HBM."""

    def __init__(self, v_template, shapedirs, posedirs, weights, J_regressor, parents, body_dof, hand_dof=0,
                 hands_mean=None, selected_components=None):
        lib = load()
        require_device()
        v_template = _f64(v_template)
        V = v_template.shape[0]
        parents = np.ascontiguousarray(parents, dtype=np.int32)
        K = parents.shape[0]
        shapedirs = _f64(shapedirs)
        if hasattr(J_regressor, 'toarray'):
            J_regressor = J_regressor.toarray()
        J_regressor = _f64(J_regressor)
        posedirs = _f64(posedirs).reshape(V, 3, -1)
        weights = _f64(weights)
        assert posedirs.shape[2] == 9 * (K - 1), (posedirs.shape, K)
        assert weights.shape == (V, K) and J_regressor.shape == (K, V)
        NB = shapedirs.shape[-1]
        d = ModelDesc()
        d.V, d.K, d.NB, d.body_dof, d.hand_dof = V, K, NB, int(body_dof), int(hand_dof)
        keep = [parents, v_template, shapedirs, posedirs, weights, J_regressor]
        d.parents = _ip(parents)
        d.v_template = _dp(v_template); d.shapedirs = _dp(shapedirs); d.posedirs = _dp(posedirs)
        d.weights = _dp(weights); d.J_regressor = _dp(J_regressor)
        if hand_dof:
            hm = _f64(hands_mean); sc = _f64(selected_components)
            assert hm.shape == (3 * K - body_dof,) and sc.shape == (hand_dof, 3 * K - body_dof)
            keep += [hm, sc]
            d.hands_mean = _dp(hm); d.selected_components = _dp(sc)
        self.V, self.K, self.NB, self.body_dof, self.hand_dof = V, K, NB, int(body_dof), int(hand_dof)
        self.P = 3 * K
        self.NP = self.body_dof + self.hand_dof
        self.handle = C.c_void_p()
        check(lib.moshii_model_create(C.byref(d), C.byref(self.handle)))
        del keep

    def set_betas(self, betas):
        b = _f64(np.ravel(betas))
        check(load().moshii_model_set_betas(self.handle, _dp(b), b.shape[0]))

    def set_free_shape(self, start, count):
        """Shapedirs columns [start, start+count) become Step-2 free variables (expression / DMPL); call before
        creating attachments."""
        check(load().moshii_model_set_free_shape(self.handle, int(start), int(count)))
        self.n_free_shape = int(count)

    def joints(self):
        out = np.zeros((self.K, 3))
        check(load().moshii_model_get_joints(self.handle, _dp(out)))
        return out

    def lbs_forward(self, pose, trans, dtype=np.float64):
        """verts[F,V,3] for pose variables pose[F,NP], trans[F,3] (host arrays)."""
        pose = np.ascontiguousarray(np.atleast_2d(pose), dtype=dtype)
        trans = np.ascontiguousarray(np.atleast_2d(trans), dtype=dtype)
        F = pose.shape[0]
        assert pose.shape == (F, self.NP) and trans.shape == (F, 3)
        out = np.zeros((F, self.V, 3), dtype=dtype)
        fn = load().moshii_lbs_forward_f64 if dtype == np.float64 else load().moshii_lbs_forward_f32
        check(fn(self.handle, F, pose.ctypes.data, trans.ctypes.data, out.ctypes.data, BUFFERS_HOST, None))
        return out

    def lbs_forward_device(self, F, pose_ptr, trans_ptr, out_ptr, stream=None, f32=True):
        fn = load().moshii_lbs_forward_f32 if f32 else load().moshii_lbs_forward_f64
        check(fn(self.handle, F, pose_ptr, trans_ptr, out_ptr, BUFFERS_DEVICE, stream))

    def close(self):
        if getattr(self, 'handle', None) is not None and self.handle.value:
            load().moshii_model_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Prior:
    """moshii_prior_t: prepared max-mixture prior (means, chols of precisions, re-normalised weights)."""

    def __init__(self, means, chols, weights):
        means = _f64(means); chols = _f64(chols); weights = _f64(np.ravel(weights))
        G, npose = means.shape
        assert chols.shape == (G, npose, npose) and weights.shape == (G,)
        self.G, self.npose = G, npose
        self.handle = C.c_void_p()
        check(load().moshii_prior_create(G, npose, _dp(means), _dp(chols), _dp(weights), C.byref(self.handle)))

    def close(self):
        if getattr(self, 'handle', None) is not None and self.handle.value:
            load().moshii_prior_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Attachment:
    """moshii_attach_t: closest[M,3] vertex ids + coef[M,3] -> compact marker-vertex model slice."""

    def __init__(self, model: Model, closest, coef):
        closest = np.ascontiguousarray(closest, dtype=np.int32)
        coef = _f64(coef)
        M = closest.shape[0]
        assert closest.shape == (M, 3) and coef.shape == (M, 3)
        self.model = model
        self.M = M
        self.handle = C.c_void_p()
        check(load().moshii_attach_create(model.handle, M, _ip(closest), _dp(coef), C.byref(self.handle)))

    def markers(self, pose, trans):
        pose = _f64(np.atleast_2d(pose)); trans = _f64(np.atleast_2d(trans))
        F = pose.shape[0]
        out = np.zeros((F, self.M, 3))
        check(load().moshii_attach_markers(self.handle, F, _dp(pose), _dp(trans), _dp(out)))
        return out

    def close(self):
        if getattr(self, 'handle', None) is not None and self.handle.value:
            load().moshii_attach_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_opts(weights, step1_ids, step2_ids, body_ids, finger_ids, maxiter=100, e3_first=1e-3, e3=1e-2,
              delta0=0.5, num_train_markers=46.0, face_ids=(), n_shape=0, shape_kind=None):
    """SolveOpts + the arrays it points to (keep the returned tuple alive during the call).
    face_ids: jaw pose ids of optimize_face (also part of step2_ids); n_shape / shape_kind ('expr' | 'dmpl'): the free
    shape block declared with Model.set_free_shape."""
    o = SolveOpts()
    o.wt_data = float(weights['stageii_wt_data']); o.wt_velo = float(weights['stageii_wt_velo'])
    o.wt_poseB = float(weights['stageii_wt_poseB']); o.wt_poseH = float(weights['stageii_wt_poseH'])
    o.wt_annealing = float(weights['stageii_wt_annealing'])
    o.num_train_markers = float(num_train_markers)
    o.e3_first, o.e3, o.delta0, o.maxiter = float(e3_first), float(e3), float(delta0), int(maxiter)
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (step1_ids, step2_ids, body_ids, finger_ids, face_ids)]
    o.n_step1, o.n_step2, o.n_body, o.n_finger, o.n_face = (len(a) for a in arrs)
    o.step1_ids, o.step2_ids, o.body_ids, o.finger_ids, o.face_ids = (_ip(a) for a in arrs)
    o.wt_poseF = float(weights.get('stageii_wt_poseF', 1.0))
    o.n_shape = int(n_shape)
    if o.n_shape:
        assert shape_kind in ('expr', 'dmpl')
        o.wt_shape = float(weights['stageii_wt_expr' if shape_kind == 'expr' else 'stageii_wt_dmpl'])
        o.wt_shape_stay = 6.0 if shape_kind == 'dmpl' else 0.0     # chmosh.py:697
    return o, arrs


def coop_group(g):
    """MOSHII_COOP_GROUP(g) of include/moshii.h: the cooperative-chain request in the `flags` of moshii_chain_solve."""
    return (int(g) & 0xff) << 8


def chain_solve_host(model: Model, prior, opts_tuple, chains, coop=0):
    """Run moshii_chain_solve on host buffers.
    chains: list of dict(attach, obs[F,M,3], vis[F,M], first=True, init_pose=None, init_trans=None, init_pose_prev=None,
    init_shape=None).  Returns list of dict(pose, fullpose, trans, markers_sim, errs[F,NERR], iters, status, shape[F,n_shape]).
    coop = g > 0: cooperative chains, g workgroups (CUs) per chain (MOSHII_COOP_GROUP(g) in the flags of the C ABI)."""
    lib = load()
    opts, _keep = opts_tuple
    n = len(chains)
    descs = (ChainDesc * n)()
    outs, keep = [], []
    for i, ch in enumerate(chains):
        att = ch['attach']
        obs = _f64(ch['obs']); vis = np.ascontiguousarray(ch['vis'], dtype=np.uint8)
        F, M = vis.shape
        assert obs.shape == (F, M, 3) and M == att.M
        o = dict(pose=np.zeros((F, model.NP)), fullpose=np.zeros((F, model.P)), trans=np.zeros((F, 3)),
                 markers_sim=np.zeros((F, M, 3)), errs=np.zeros((F, NERR)), iters=np.zeros((F, 2), dtype=np.int32),
                 status=np.zeros(F, dtype=np.int32), shape=np.zeros((F, int(opts.n_shape))))
        d = descs[i]
        d.attach = att.handle
        d.F = F
        d.first_frame_schedule = 1 if ch.get('first', True) else 0
        d.obs = obs.ctypes.data
        d.vis = vis.ctypes.data
        keep += [obs, vis]
        for key, fld in (('init_pose', 'init_pose'), ('init_trans', 'init_trans'), ('init_pose_prev', 'init_pose_prev'),
                         ('init_shape', 'init_shape')):
            if ch.get(key) is not None:
                a = _f64(ch[key]); keep.append(a)
                setattr(d, fld, _dp(a))
        for key in ('pose', 'fullpose', 'trans', 'markers_sim', 'errs', 'iters', 'status'):
            setattr(d, key, o[key].ctypes.data)
        if opts.n_shape:
            d.shape = o['shape'].ctypes.data
        outs.append(o)
    check(lib.moshii_chain_solve(model.handle, prior.handle if prior is not None else None, C.byref(opts), n, descs,
                                 BUFFERS_HOST | coop_group(coop), None))
    del keep
    return outs


def plan_chunks(F, num_chunks, warmup, cap=1 << 20):
    """moshii_plan_chunks -> (starts, launch_starts) int32 arrays (host arithmetic only; works without a GPU)."""
    n = max(1, int(num_chunks))
    starts = np.zeros(n, dtype=np.int32); launch = np.zeros(n, dtype=np.int32)
    c = load().moshii_plan_chunks(int(F), n, int(warmup), int(cap), _ip(starts), _ip(launch))
    if c < 0:
        check(c)
    return starts[:c].copy(), launch[:c].copy()


def sequence_solve_host(model: Model, prior, opts_tuple, seqs, num_chunks=0, warmup=32, verify_tol=1e-11, coop=0):
    """moshii_sequence_solve on host buffers.  seqs: list of dict(attach, obs[F,M,3], vis[F,M], init_pose=None,
    init_trans=None, init_pose_prev=None) -- with init_* the sequence continues a chain from that state instead of running
    the first-frame schedule.  Returns (list of per-sequence output dicts as chain_solve_host, report dict)."""
    lib = load()
    opts, _keep = opts_tuple
    n = len(seqs)
    descs = (SequenceDesc * n)()
    outs, keep = [], []
    for i, sq in enumerate(seqs):
        att = sq['attach']
        obs = _f64(sq['obs']); vis = np.ascontiguousarray(sq['vis'], dtype=np.uint8)
        F, M = vis.shape
        assert obs.shape == (F, M, 3) and M == att.M
        o = dict(pose=np.zeros((F, model.NP)), fullpose=np.zeros((F, model.P)), trans=np.zeros((F, 3)),
                 markers_sim=np.zeros((F, M, 3)), errs=np.zeros((F, NERR)), iters=np.zeros((F, 2), dtype=np.int32),
                 status=np.zeros(F, dtype=np.int32), shape=np.zeros((F, int(opts.n_shape))))
        d = descs[i]
        d.attach = att.handle; d.F = F; d.obs = obs.ctypes.data; d.vis = vis.ctypes.data
        keep += [obs, vis]
        for key in ('init_pose', 'init_trans', 'init_pose_prev', 'init_shape'):
            if sq.get(key) is not None:
                a = _f64(sq[key]); keep.append(a)
                setattr(d, key, _dp(a))
        for key in ('pose', 'fullpose', 'trans', 'markers_sim', 'errs', 'iters', 'status'):
            setattr(d, key, o[key].ctypes.data)
        if int(opts.n_shape):
            d.shape = o['shape'].ctypes.data
        outs.append(o)
    co = ChunkOpts(int(num_chunks), int(warmup), float(verify_tol))
    rep = ChunkReport()
    check(lib.moshii_sequence_solve(model.handle, prior.handle if prior is not None else None, C.byref(opts), n, descs,
                                    C.byref(co), BUFFERS_HOST | coop_group(coop), None, C.byref(rep)))
    del keep
    report = {k: getattr(rep, k) for k, _ in ChunkReport._fields_}
    return outs, report


def last_launch_info():
    name = C.create_string_buffer(128)
    lds = C.c_int32(0); thr = C.c_int32(0)
    load().moshii_last_launch_info(name, 128, C.byref(lds), C.byref(thr))
    return name.value.decode(), lds.value, thr.value


# ---- Stage-I (moshii_stagei_solve) -------------------------------------------------------------------------------
class StageIDesc(C.Structure):
    _fields_ = [('n_frames', C.c_int32), ('M', C.c_int32), ('n_faces', C.c_int32), ('nb', C.c_int32),
                ('faces', C.c_void_p), ('marker_vids', C.c_void_p), ('m2b', C.c_void_p), ('wt_init', C.c_void_p),
                ('n_obs', C.c_void_p), ('obs_ids', C.c_void_p), ('obs', C.c_void_p),
                ('exclude_vids', C.c_void_p), ('n_exclude', C.c_int32), ('betas_init', C.c_void_p),
                ('wt_data', C.c_double), ('wt_poseB', C.c_double), ('wt_poseH', C.c_double), ('wt_betas', C.c_double),
                ('wt_surf', C.c_double), ('annealing', C.c_void_p), ('n_anneal', C.c_int32),
                ('pose_ids', C.c_void_p), ('n_pose_ids', C.c_int32), ('body_ids', C.c_void_p), ('n_body', C.c_int32),
                ('finger_ids', C.c_void_p), ('n_finger', C.c_int32),
                ('n_expr', C.c_int32), ('expr_start', C.c_int32), ('face_ids', C.c_void_p), ('n_face', C.c_int32),
                ('wt_expr', C.c_double), ('wt_poseF', C.c_double),
                ('head_ids', C.c_void_p), ('head_corr', C.c_void_p), ('n_head', C.c_int32), ('n_head_rows', C.c_int32),
                ('wt_init_head', C.c_double), ('maxiter', C.c_int32), ('stagei_lr', C.c_double),
                ('sharded', C.c_int32), ('frame_lo', C.c_int32), ('frame_hi', C.c_int32), ('owns_shared_rows', C.c_int32),
                ('allreduce_sum', C.c_void_p), ('allreduce_user', C.c_void_p),
                ('betas', C.c_void_p), ('markers_latent', C.c_void_p), ('markers_latent_vids', C.c_void_p),
                ('pose', C.c_void_p), ('trans', C.c_void_p), ('markers_sim', C.c_void_p), ('expression', C.c_void_p),
                ('errs', C.c_void_p), ('iters', C.c_void_p), ('extra_initial_rigid_adjustment', C.c_int32),
                ('allreduce_on_device', C.c_int32), ('init_sq', C.c_void_p)]


ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_void_p)
STAGEI_ERR_NAMES = ('data', 'poseB', 'init', 'beta', 'surf', 'poseH', 'init_head_corr', 'poseF')   # 'beta' holds expr with n_expr > 0


def stagei_desc(NP, faces, marker_vids, m2b, wt_init, frames, nb, weights, pose_ids, body_ids, finger_ids=(), exclude_vids=None,
                betas_init=None, maxiter=100, stagei_lr=1e-3, head_corr=None, wt_init_head=None, frame_range=None,
                owns_shared_rows=True, allreduce=None, n_expr=0, expr_start=0, face_ids=(), extra_initial_rigid_adjustment=False,
                allreduce_on_device=False):
    """Fill a StageIDesc from NumPy data.  `frames`: list of (latent ids, obs[n,3]).  Returns (desc, outputs dict, keep-alive list)."""
    keep = []

    def ptr(a, dt):
        a = np.ascontiguousarray(a, dtype=dt); keep.append(a)
        return a.ctypes.data if a.size else None
    F, M = len(frames), len(marker_vids)
    d = StageIDesc()
    d.n_frames, d.M, d.n_faces, d.nb = F, M, len(faces), int(nb)
    d.faces = ptr(faces, np.int32); d.marker_vids = ptr(marker_vids, np.int32)
    d.m2b = ptr(m2b, np.float64); d.wt_init = ptr(wt_init, np.float64)
    d.n_obs = ptr([len(ids) for ids, _ in frames], np.int32)
    d.obs_ids = ptr(np.concatenate([np.asarray(ids) for ids, _ in frames]), np.int32)
    d.obs = ptr(np.vstack([np.asarray(o, dtype=np.float64).reshape(-1, 3) for _, o in frames]), np.float64)
    ex = np.zeros(0, np.int32) if exclude_vids is None else np.asarray(exclude_vids, np.int32)
    d.exclude_vids = ptr(ex, np.int32); d.n_exclude = len(ex)
    d.betas_init = ptr(np.asarray(betas_init, np.float64)[:nb], np.float64) if betas_init is not None else None
    d.wt_data, d.wt_poseB, d.wt_poseH = weights['stagei_wt_data'], weights['stagei_wt_poseB'], weights['stagei_wt_poseH']
    d.wt_betas, d.wt_surf = weights['stagei_wt_betas'], weights['stagei_wt_surf']
    ann = list(weights['stagei_wt_annealing'])
    d.annealing = ptr(ann, np.float64); d.n_anneal = len(ann)
    d.pose_ids = ptr(pose_ids, np.int32); d.n_pose_ids = len(pose_ids)
    d.body_ids = ptr(body_ids, np.int32); d.n_body = len(body_ids)
    d.finger_ids = ptr(list(finger_ids), np.int32); d.n_finger = len(finger_ids)
    d.maxiter, d.stagei_lr = int(maxiter), float(stagei_lr)
    d.extra_initial_rigid_adjustment = 1 if extra_initial_rigid_adjustment else 0
    d.n_expr, d.expr_start = int(n_expr), int(expr_start)
    d.face_ids = ptr(list(face_ids), np.int32); d.n_face = len(face_ids)
    d.wt_expr, d.wt_poseF = float(weights.get('stagei_wt_expr', 0.0)), float(weights.get('stagei_wt_poseF', 0.0))
    if head_corr is not None:
        hid, Cm = head_corr
        Cm = np.atleast_2d(np.asarray(Cm, np.float64))
        assert Cm.shape[1] == len(hid)
        d.head_ids = ptr(hid, np.int32); d.head_corr = ptr(Cm, np.float64); d.n_head, d.n_head_rows = Cm.shape[1], Cm.shape[0]
    d.wt_init_head = float(weights['stagei_wt_init'] if wt_init_head is None else wt_init_head)
    if allreduce is not None:
        # allreduce(array) sums a 1-D float64 NumPy array in place over the ranks (moshpp_amd.parallel.make_allreduce); with
        # allreduce_on_device it is allreduce(device_pointer, count) instead (parallel.make_allreduce_device: RCCL on the solver's
        # own buffers)
        def _cb(buf, count, _user):
            try:
                if allreduce_on_device:
                    allreduce(C.cast(buf, C.c_void_p).value, int(count))
                else:
                    allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:          # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLREDUCE_CB(_cb)
        keep.append(cb)
        d.sharded = 1
        d.allreduce_on_device = 1 if allreduce_on_device else 0
        d.frame_lo, d.frame_hi = (0, F) if frame_range is None else (int(frame_range[0]), int(frame_range[1]))
        d.owns_shared_rows = 1 if owns_shared_rows else 0
        d.allreduce_sum = C.cast(cb, C.c_void_p)
    out = dict(betas=np.zeros(max(nb, 1)), markers_latent=np.zeros((M, 3)), markers_latent_vids=np.zeros(M, np.int32),
               pose=np.zeros((F, NP)), trans=np.zeros((F, 3)), markers_sim=np.zeros((F, M, 3)), expression=np.zeros((F, max(int(n_expr), 1))), errs=np.zeros(8),
               iters=np.zeros(1, np.int32), init_sq=np.zeros(M))
    for k, v in out.items():
        setattr(d, k, v.ctypes.data)
    out['betas'] = out['betas'][:nb]
    out['expression'] = out['expression'][:, :int(n_expr)]
    keep.append(out)
    return d, out, keep


EXPORTS['moshii_stagei_solve'] = (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(StageIDesc), C.c_void_p])


def stagei_solve_host(model: Model, prior, **kw):
    """moshii_stagei_solve on host buffers; kw as stagei_desc().  Returns dict(betas, markers_latent, markers_latent_vids, pose,
    trans, errs{term: SSE}, iters, init_sq[M]: every marker's share of errs['init'])."""
    require_device()
    desc, out, _keep = stagei_desc(NP=model.NP, **kw)
    check(load().moshii_stagei_solve(model.handle, prior.handle if prior is not None else None, C.byref(desc), None))
    out = dict(out)
    out['errs'] = dict(zip(STAGEI_ERR_NAMES, out['errs'].tolist()))
    out['iters'] = int(out['iters'][0])
    return out
