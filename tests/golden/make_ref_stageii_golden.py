"""Generates tests/golden/ref_stageii.npz by EXECUTING the reference's own `mosh_stageii`
(/root/reference/src/moshpp/chmosh.py:458-741, function source taken from the file, unmodified) on seeded synthetic
inputs, together with the reference code it drives:

  * models.bodymodel_loader.load_moshpp_models (+ AliasedBetas)            bodymodel_loader.py:52-153
  * models.smpl_fast_derivatives.load_surface_model, SmplModelLBS.__init__ smpl_fast_derivatives.py:52-244
        (shaped template, regressed joints, hand-PCA fullpose expression -- the reference's statements)
  * transformed_lm.TransformedCoeffs / TransformedLms                       transformed_lm.py:45-162
  * prior.gmm_prior_ch.create_gmm_body_prior / MaxMixtureComplete           gmm_prior_ch.py:42-134
  * rigid_transformations.perform_rigid_adjustment / rigid_landmark_transform :39-83
  * tools.mocap_interface.MocapSession                                      mocap_interface.py:87-279

so that the Stage-II SCHEDULE -- which residual blocks exist in which solve, their weights and annealing, which variables
are free in the first-frame rounds / Step 1 / Step 2, when pose_prev is refreshed relative to the velocity term, what is
recorded per frame and under which keys -- is the reference's executed code, not a restatement.

chumpy, psbody.smpl, cv2, loguru and omegaconf are not installable here.  What stands in for them (and therefore stays
"restated", see oracle/stageii_oracle.py's header):

  * `chumpy`: the LAZY stand-in below -- expression nodes that re-evaluate when a variable they depend on changes, `Ch`
    with chumpy's dterm / on_changed(which) / compute_r protocol, item assignment on variables.  No automatic
    differentiation.
  * `ch.minimize(method='dogleg')`: the oracle's minimize_dogleg (the restated chumpy control flow) on the residual
    vector the REFERENCE built, its Jacobian taken by central differences of that residual (h = 1e-6) -- so neither the
    oracle's residual code nor its analytic Jacobian takes part in producing this fixture.
  * `psbody.smpl.verts.verts_decorated(...)`.r: the oracle's LBS forward (verts_forward) on the arrays SmplModelLBS hands
    over; `cv2.Rodrigues`: the oracle's rotmat_to_rotvec.

Run in the build container only (needs /root/reference); the npz is committed.  tests/test_ref_golden.py regenerates the
inputs from the seeds recorded here and holds the oracle chain (and, on a GPU, the kernel) to the recorded trajectory.
"""
import ast
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/src/moshpp'

from oracle import stageii_oracle as so  # noqa: E402

# ---------------------------------------------------------------------------------------------------
# lazy chumpy stand-in
# ---------------------------------------------------------------------------------------------------
_CLOCK = [1]


def _tick():
    _CLOCK[0] += 1
    return _CLOCK[0]


def _is_ch(x):
    return isinstance(x, Ch)


def _val(x):
    return x.r if _is_ch(x) else np.asarray(x)


class Ch(object):
    """chumpy.Ch protocol: class attributes `dterms` / `terms` name the inputs; keyword arguments that name one are set
    before __init__ runs; r -> [on_changed(which) if any input changed since the last evaluation] -> compute_r()."""
    dterms = ()
    terms = ()

    def __new__(cls, *args, **kwargs):
        obj = object.__new__(cls)
        d = obj.__dict__
        d['_dirty'] = set(); d['_seen'] = {}; d['_cache'] = None; d['_cache_ver'] = -1; d['_own_ver'] = _tick()
        d['_ver_memo'] = (-1, 0); d['_extra'] = []; d['_fn'] = None
        if cls is Ch and len(args) == 1 and callable(args[0]) and not _is_ch(args[0]):
            # chumpy: Ch(lambda a, b: expression) -- a node whose inputs are the lambda's arguments, assigned as attributes afterwards
            import inspect
            d['_fn'] = args[0]
            d['_extra'] = list(inspect.signature(args[0]).parameters)
            return obj
        if args and cls.__init__ is Ch.__init__:
            # positional construction of a declared node (TriEdges(f, 1, 0, v), CrossProduct(a, b)): term_order, else terms + dterms
            order = getattr(cls, 'term_order', None)
            if order is None:
                order = []
                for t in (cls.terms, cls.dterms):
                    order += [t] if isinstance(t, str) else list(t)
            for k, v in zip(order, args):
                setattr(obj, k, v)
        for k, v in kwargs.items():
            if k in obj._input_names():
                setattr(obj, k, v)
        return obj

    def __init__(self, *args, **kwargs):
        pass

    def _input_names(self):
        names = []
        for t in (type(self).dterms, type(self).terms):
            names += [t] if isinstance(t, str) else list(t)
        return names + self.__dict__.get('_extra', [])

    def _dterm_names(self):
        t = type(self).dterms
        return ([t] if isinstance(t, str) else list(t)) + self.__dict__.get('_extra', [])

    def __setattr__(self, name, value):
        if name in self._input_names():
            if name in self._dterm_names() and not _is_ch(value) and isinstance(value, (np.ndarray, list, tuple, float, int)):
                value = array(value)
            self.__dict__['_dirty'].add(name)
            self.__dict__['_own_ver'] = _tick()
        object.__setattr__(self, name, value)

    def add_dterm(self, name, value):
        if name not in self._input_names():
            self.__dict__['_extra'].append(name)
        setattr(self, name, value)

    # -- evaluation ------------------------------------------------------------------------------
    def _children(self):
        return [(n, getattr(self, n)) for n in self._input_names() if _is_ch(self.__dict__.get(n))]

    def _version(self):
        clk, v = self.__dict__['_ver_memo']
        if clk == _CLOCK[0]:
            return v
        v = self.__dict__['_own_ver']
        for _, c in self._children():
            v = max(v, c._version())
        self.__dict__['_ver_memo'] = (_CLOCK[0], v)
        return v

    def on_changed(self, which):
        pass

    def compute_r(self):
        fn = self.__dict__.get('_fn')
        if fn is None:
            raise NotImplementedError
        return _val(fn(**{n: getattr(self, n) for n in self.__dict__['_extra']}))

    @property
    def r(self):
        d = self.__dict__
        if d['_cache'] is None or self._version() != d['_cache_ver']:
            dirty = set(d['_dirty'])
            for n, c in self._children():
                cv = c._version()
                if d['_seen'].get(n) != cv:
                    dirty.add(n)
            d['_dirty'] = set()
            if dirty:
                self.on_changed(sorted(dirty))
                d['_dirty'] = set()
            for n, c in self._children():
                d['_seen'][n] = c._version()
            val = np.array(_val(self.compute_r()), dtype=np.float64, copy=True)
            d['_cache'] = val
            d['_cache_ver'] = self._version()
        return d['_cache']

    # -- ndarray-like surface --------------------------------------------------------------------
    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.r, dtype=dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        # `numpy_scalar * node` etc. must stay lazy (chumpy: __array_priority__); other ufuncs (np.isnan, np.log) take values
        if method == '__call__' and not kwargs and ufunc in (np.add, np.subtract, np.multiply, np.divide, np.power):
            return Op(ufunc, *inputs)
        return getattr(ufunc, method)(*[_val(i) for i in inputs], **kwargs)

    shape = property(lambda self: self.r.shape)
    size = property(lambda self: self.r.size)
    ndim = property(lambda self: self.r.ndim)
    T = property(lambda self: Op(lambda a: a.T, self))

    def __len__(self):
        return len(self.r)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __getitem__(self, idx):
        return Select(self, idx)

    def __add__(self, o): return Op(np.add, self, o)
    def __radd__(self, o): return Op(np.add, o, self)
    def __sub__(self, o): return Op(np.subtract, self, o)
    def __rsub__(self, o): return Op(np.subtract, o, self)
    def __mul__(self, o): return Op(np.multiply, self, o)
    def __rmul__(self, o): return Op(np.multiply, o, self)
    def __truediv__(self, o): return Op(np.divide, self, o)
    def __rtruediv__(self, o): return Op(np.divide, o, self)
    def __pow__(self, o): return Op(np.power, self, o)
    def __neg__(self): return Op(np.negative, self)
    def dot(self, o): return Op(np.dot, self, o)
    def reshape(self, *shape): return Op(lambda a: a.reshape(*shape), self)
    def ravel(self): return Op(np.ravel, self)
    def copy(self): return array(self.r.copy())
    def sum(self, axis=None): return Op(lambda a: np.atleast_1d(a.sum(axis=axis)), self)   # chumpy: sum() has shape (1,)


class Op(Ch):
    """fn(*args) with args constants or nodes."""

    def __init__(self, fn, *args):
        d = self.__dict__
        d['fn'] = fn; d['args'] = args

    def _children(self):
        return [(str(i), a) for i, a in enumerate(self.__dict__['args']) if _is_ch(a)]

    def compute_r(self):
        return self.__dict__['fn'](*[_val(a) for a in self.__dict__['args']])


class Array(Ch):
    """A variable / constant array (ch.array, ch.zeros)."""

    def __init__(self, x):
        self.__dict__['x'] = np.array(x, dtype=np.float64, copy=True)

    def _children(self):
        return []

    @property
    def r(self):
        return self.__dict__['x']

    def __setitem__(self, idx, value):
        self.__dict__['x'][idx] = _val(value)
        self.__dict__['_own_ver'] = _tick()

    def set_value(self, v):
        self[:] = np.asarray(v).reshape(self.__dict__['x'].shape)


class Select(Ch):
    def __init__(self, parent, idx):
        d = self.__dict__
        d['parent'] = parent; d['idx'] = idx

    def _children(self):
        return [('parent', self.__dict__['parent'])]

    def compute_r(self):
        out = self.__dict__['parent'].r[self.__dict__['idx']]
        return np.atleast_1d(out) if np.ndim(out) == 0 else out   # chumpy: x[i] of a vector is a 1-element Ch

    def _base_and_map(self):
        """(base Array, flat indices of this view's elements in it): chumpy's Select is a Permute, and item assignment on a
        Permute of a leaf writes through to the leaf (ch.py: Ch.__setitem__)."""
        p = self.__dict__['parent']
        if isinstance(p, Array):
            base, pmap = p, np.arange(p.r.size).reshape(p.r.shape)
        else:
            assert isinstance(p, Select), 'assignment through a view: only views of arrays'
            base, pmap = p._base_and_map()
        return base, pmap[self.__dict__['idx']]

    def __setitem__(self, idx, value):       # can_model.shapedirs[:, :, nb:nb + nd] = dmpl_pcs (chmosh.py:512): a view of the loaded model's array
        base, fmap = self._base_and_map()
        x = base.__dict__['x']
        x.reshape(-1)[np.asarray(fmap[idx]).ravel()] = np.broadcast_to(_val(value), np.shape(fmap[idx])).ravel()
        base.__dict__['_own_ver'] = _tick()

    def set_value(self, v):
        p = self.__dict__['parent']
        assert isinstance(p, Array), 'free variables are arrays or index views of arrays'
        p[self.__dict__['idx']] = np.asarray(v).reshape(np.shape(p.r[self.__dict__['idx']]))


def array(x):
    return x if _is_ch(x) else Array(x)


class MatVecMult(Ch):
    def __init__(self, mtx, vec):
        self.__dict__['mtx'] = mtx; self.__dict__['vec'] = vec

    def _children(self):
        return [('vec', self.__dict__['vec'])]

    def compute_r(self):
        return np.asarray(self.__dict__['mtx'].dot(_val(self.__dict__['vec']))).ravel()


N_MINIMIZE = []   # (n free variables, n residual rows, dogleg iterations, residual evaluations) per ch.minimize call


def minimize(fun, x0, method='dogleg', options=None, **kw):
    """ch.minimize(method='dogleg'): the oracle's dogleg on the residuals the caller built; Jacobian by central differences."""
    assert method == 'dogleg'
    objs = list(fun.values()) if isinstance(fun, dict) else list(fun)
    free = list(x0)
    sizes = [int(np.size(f.r)) for f in free]

    def getx():
        return np.concatenate([np.ravel(f.r) for f in free]).astype(np.float64)

    def setx(x):
        o = 0
        for f, n in zip(free, sizes):
            f.set_value(x[o:o + n]); o += n

    class Obj:
        @staticmethod
        def r(x):
            setx(x)
            return np.concatenate([np.ravel(o.r) for o in objs])

        @staticmethod
        def J(x):
            h = 1e-6
            cols = []
            for i in range(len(x)):
                xp = x.copy(); xp[i] += h
                xm = x.copy(); xm[i] -= h
                cols.append((Obj.r(xp) - Obj.r(xm)) / (2 * h))
            setx(x)
            return np.array(cols).T

    stats = {}
    x = so.minimize_dogleg(Obj, getx(), e_3=options.get('e_3', 0.0), delta_0=options.get('delta_0'),
                           maxiter=options.get('maxiter', 100), stats=stats)
    setx(x)
    N_MINIMIZE.append((len(x), len(Obj.r(x)), stats.get('iterations', -1), stats.get('n_fev', -1)))


ch = types.ModuleType('chumpy')
ch.Ch = Ch
ch.array = array
ch.asarray = array
ch.zeros = lambda n: Array(np.zeros(n))
ch.vstack = lambda xs: Op(lambda *a: np.vstack(a), *list(xs))
ch.hstack = lambda xs: Op(lambda *a: np.hstack(a), *list(xs))
ch.concatenate = lambda xs, axis=0: Op(lambda *a: np.concatenate([np.atleast_1d(v) for v in a], axis=axis), *list(xs))
ch.cross = lambda a, b: Op(np.cross, a, b)
ch.sqrt = lambda a: Op(np.sqrt, a)
ch.sum = lambda a, axis=None: Op(lambda v: np.atleast_1d(v.sum(axis=axis)), a)
ch.minimize = minimize
def depends_on(*_names):
    """chumpy.depends_on: a cached property invalidated when the named inputs change; here simply re-evaluated at every access."""
    return lambda fn: property(fn)


ch.depends_on = depends_on
ch.abs = lambda a: Op(np.abs, a)
ch_ch = types.ModuleType('chumpy.ch')
ch_ch.MatVecMult = MatVecMult
ch_ch.Ch = Ch
ch_ch.depends_on = depends_on
ch.ch = ch_ch
ch_utils = types.ModuleType('chumpy.utils')
ch_utils.row = lambda a: np.asarray(a).reshape((1, -1))
ch_utils.col = lambda a: np.asarray(a).reshape((-1, 1))
ch.utils = ch_utils
sys.modules['chumpy'] = ch
sys.modules['chumpy.ch'] = ch_ch
sys.modules['chumpy.utils'] = ch_utils


# ---------------------------------------------------------------------------------------------------
# the other absent modules
# ---------------------------------------------------------------------------------------------------
def _module(name, **attrs):
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


_quiet = types.SimpleNamespace(error=lambda *a, **k: None, info=lambda *a, **k: None, debug=lambda *a, **k: None,
                               warning=lambda *a, **k: None, success=lambda *a, **k: None)
_module('loguru', logger=_quiet)
_module('omegaconf', DictConfig=dict, OmegaConf=None)
_module('cv2', Rodrigues=lambda R: (so.rotmat_to_rotvec(np.asarray(R)).reshape(3, 1), None))
_module('human_body_prior')
_module('human_body_prior.tools')
_module('human_body_prior.tools.omni_tools', get_support_data_dir=lambda _f: '/root/reference/support_data',
        flatten_list=lambda l: [x for s in l for x in s])
_module('human_body_prior.tools.rotation_tools', rotate_points_xyz=None)
for _n, _a in {'ezc3d': {}, 'body_visualizer': {}, 'body_visualizer.mesh': {}, 'body_visualizer.tools': {},
               'body_visualizer.mesh.psbody_mesh_sphere': {'points_to_spheres': None},
               'body_visualizer.tools.vis_tools': {'colors': {}}, 'psbody': {}, 'psbody.mesh': {'Mesh': None},
               'psbody.mesh.meshviewer': {'MeshViewer': None}, 'psbody.mesh.sphere': {'Sphere': None},
               'psbody.smpl': {}, 'psbody.smpl.fast_derivatives': {},
               'psbody.smpl.fast_derivatives.smplcpp_chumpy': {'lbs_derivatives_wrt_pose': None, 'lbs_derivatives_wrt_shape': None}}.items():
    _module(_n, **_a)


class VertsDecorated(Ch):
    """psbody.smpl.verts.verts_decorated(...) stand-in: `.r` = the oracle's LBS forward of the arrays SmplModelLBS passes in
    (pose = the reference's own fullpose expression); the attributes SmplModelLBS copies from it are passed through."""
    dterms = 'trans', 'pose', 'betas'

    def __init__(self, trans, pose, v_template, J, weights, kintree_table, bs_style, f, bs_type, posedirs, betas, shapedirs,
                 want_Jtr=True):
        d = self.__dict__
        d.update(v_template=v_template, J=J, weights=weights, kintree_table=kintree_table, bs_style=bs_style, f=f, bs_type=bs_type,
                 posedirs=posedirs, shapedirs=shapedirs, v_shaped=None, A=None, A_global=None, Jtr=None, v_posed=None,
                 A_weighted=None, J_regressor=None, _prepared=None)

    def _model(self):
        d = self.__dict__
        b = np.asarray(self.betas.r, dtype=np.float64)
        sv = d['shapedirs']._version() if _is_ch(d['shapedirs']) else 0     # (the DMPL branch rewrites shapedirs columns after construction)
        if d['_prepared'] is None or not np.array_equal(d['_prepared'][0], b) or d['_prepared'][2] != sv:
            kt = np.asarray(d['kintree_table'])
            parents = [-1] + [int(p) for p in kt[0, 1:]]
            nb = len(b)
            model = dict(v_template=_val(d['v_template']), shapedirs=_val(d['shapedirs'])[:, :, :nb], posedirs=_val(d['posedirs']),
                         weights=_val(d['weights']), J_regressor=d['J_regressor'], parents=parents,
                         body_dof=3 * kt.shape[1], hand_dof=0, hands_mean=None, selected_components=None)
            d['_prepared'] = (b.copy(), so.prepare_model(model, b), sv)
        return d['_prepared'][1]

    def compute_r(self):
        return so.verts_forward(self._model(), np.asarray(self.pose.r, dtype=np.float64), np.asarray(self.trans.r, dtype=np.float64))


_module('psbody.smpl.verts', verts_decorated=VertsDecorated)


def load_ref(name, rel):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class Cfg(dict):
    """omegaconf.DictConfig stand-in: attribute and item access, item assignment."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    @staticmethod
    def of(d):
        return Cfg({k: Cfg.of(v) if isinstance(v, dict) else v for k, v in d.items()})


def run_reference_stageii(model_type, n_frames, n_markers, seed, n_verts, empty_frames=(), optimize_fingers=False,
                          optimize_toes=False, optimize_face=False, optimize_dynamics=False, n_free_shape=0):
    """Builds the seeded case (tests/golden/ref_inputs.stageii_case: files on disk, as the reference reads them) and runs the
    reference's mosh_stageii on it.  Returns (result dict, case)."""
    from tests.golden.ref_inputs import stageii_case
    tmp = tempfile.mkdtemp(prefix='ref_stageii_')
    case = stageii_case(model_type, n_frames, n_markers, seed, n_verts, tmp, empty_frames=empty_frames, finger_markers=optimize_fingers,
                        face_markers=optimize_face, n_free_shape=n_free_shape,
                        shape_kind='expr' if optimize_face else ('dmpl' if optimize_dynamics else None))
    # the reference modules, from their files
    _module('moshpp'); _module('moshpp.models'); _module('moshpp.prior'); _module('moshpp.tools'); _module('moshpp.marker_layout')
    sfd = load_ref('moshpp.models.smpl_fast_derivatives', 'models/smpl_fast_derivatives.py')
    load_ref('moshpp.prior.gmm_prior_ch', 'prior/gmm_prior_ch.py')
    bml = load_ref('moshpp.models.bodymodel_loader', 'models/bodymodel_loader.py')
    tlm = load_ref('moshpp.transformed_lm', 'transformed_lm.py')
    rig = load_ref('moshpp.rigid_transformations', 'rigid_transformations.py')
    mi = load_ref('moshpp.tools.mocap_interface', 'tools/mocap_interface.py')
    lm = load_ref('moshpp.marker_layout.labels_map', 'marker_layout/labels_map.py')
    # SmplModelLBS does not hand J_regressor to verts_decorated before `_inner_model.J_regressor = ...` at the end of its
    # constructor (smpl_fast_derivatives.py:241); the stand-in reads it from there.
    src = open(os.path.join(REF, 'chmosh.py')).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'mosh_stageii'][0]
    ns = {'np': np, 'ch': ch, 'pickle': pickle, 'DictConfig': dict, 'logger': _quiet, 'MocapSession': mi.MocapSession,
          'general_labels_map': lm.general_labels_map, 'load_moshpp_models': bml.load_moshpp_models,
          'TransformedCoeffs': tlm.TransformedCoeffs, 'TransformedLms': tlm.TransformedLms,
          'perform_rigid_adjustment': rig.perform_rigid_adjustment, 'visualize_pose_estimate': None,
          # the DMPL branch reads its pickle through a TEXT-mode open() (chmosh.py:511, Python-2 era): pickle.load needs bytes
          'open': lambda fname, *a: open(fname, 'rb')}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'chmosh.py', 'exec'), ns)
    cfg = Cfg.of(dict(
        mocap=dict(unit='m', rotate=None, subject_name=None, multi_subject=False, start_fidx=0, end_fidx=-1, ds_rate=1),
        moshpp=dict(optimize_fingers=optimize_fingers, optimize_face=optimize_face, optimize_toes=optimize_toes, optimize_dynamics=optimize_dynamics,
                    pose_hand_prior_fname=case['hand_prior_fname'], pose_body_prior_fname=case['body_prior_fname'], verbosity=0,
                    visualization=dict(marker_radius=dict(body=0.009))),
        surface_model=dict(fname=case['model_fname'], type=model_type, use_hands_mean=case['use_hands_mean'],
                           dof_per_hand=case['dof_per_hand'], num_betas=16, num_dmpls=n_free_shape if optimize_dynamics else 0,
                           num_expressions=n_free_shape if optimize_face else 0, betas_expr_start_id=16, dmpl_fname=case['dmpl_fname']),
        opt_settings=dict(maxiter=100, weights=dict(so.stageii_weights_default()))))
    del N_MINIMIZE[:]
    out = ns['mosh_stageii'](case['mocap_fname'], cfg, case['markers_latent'], case['latent_labels'], case['betas'],
                             case['marker_meta'])
    return out, case


CASES = {   # name: dict(model_type, frames, markers, seed, vertices, + switches)
    'smplh_body': dict(mt='smplh', F=6, M=53, seed=3, V=1500, empty=(3,)),   # BASELINE config 2's shape: SMPL-H, 53 markers, fixed betas; one empty frame
    'smpl_body': dict(mt='smpl', F=5, M=41, seed=4, V=1200),                 # BASELINE config 1's shape: SMPL, 41 markers; dropouts -> annealed weights
    'smplh_fingers': dict(mt='smplh', F=4, M=66, seed=5, V=1500, fingers=True),   # Step 2 frees the hand coefficients and adds the poseH term (chmosh.py:681-683)
    # round 3: the remaining branches of the schedule
    'smplh_toes': dict(mt='smplh', F=4, M=53, seed=6, V=1500, toes=True),    # optimize_toes: pose ids 30:36 stay free (:646-647, 666-667, 678-679)
    'mano_fingers': dict(mt='mano', F=4, M=24, seed=7, V=700, fingers=True),  # MANO: no body ids, no poseB term, pose_finger_ids = all_pose_ids[3:] (:569-570)
    'smplx_face': dict(mt='smplx', F=4, M=60, seed=8, V=1600, face=True, E=4),   # jaw ids 66:69 + poseF + expr terms, expression block free (:562-567, 685-689, 721-724)
    'smplh_dmpl': dict(mt='smplh', F=5, M=53, seed=9, V=1500, dynamics=True, E=3),   # DMPL block free, dmpl / extrap_dmpl terms, dmpl_prev refresh order (:507-514, 658-659, 694-699, 719-720)
}


def main():
    out = {}
    for name, cs in CASES.items():
        mt, F, M, seed, V = cs['mt'], cs['F'], cs['M'], cs['seed'], cs['V']
        empty, fingers = cs.get('empty', ()), cs.get('fingers', False)
        res, case = run_reference_stageii(mt, F, M, seed, V, empty_frames=empty, optimize_fingers=fingers, optimize_toes=cs.get('toes', False),
                                          optimize_face=cs.get('face', False), optimize_dynamics=cs.get('dynamics', False),
                                          n_free_shape=cs.get('E', 0))
        dbg = res['stageii_debug_details']
        out[f'{name}_args'] = np.array([F, M, seed, V] + list(empty), dtype=np.int64)
        out[f'{name}_fingers'] = np.array(bool(fingers))
        out[f'{name}_switches'] = np.array([int(cs.get('toes', False)), int(cs.get('face', False)), int(cs.get('dynamics', False)), int(cs.get('E', 0))])
        for k in ('expression', 'dmpls'):
            if k in res:
                out[f'{name}_{k}'] = np.asarray(res[k])
        out[f'{name}_fullpose'] = np.asarray(res['fullpose'])
        out[f'{name}_trans'] = np.asarray(res['trans'])
        out[f'{name}_keys'] = np.array(sorted(res.keys()))
        out[f'{name}_debug_keys'] = np.array(sorted(dbg.keys()))
        out[f'{name}_err_keys'] = np.array(list(dbg['stageii_errs'].keys()))
        for k, v in dbg['stageii_errs'].items():
            out[f'{name}_err_{k}'] = np.asarray(v)
        out[f'{name}_n_obs'] = np.array([len(l) for l in dbg['labels_obs']])
        out[f'{name}_labels_obs'] = np.array(['|'.join(l) for l in dbg['labels_obs']])
        out[f'{name}_markers_sim0'] = np.asarray(dbg['markers_sim'][0])
        out[f'{name}_minimize_calls'] = np.array(N_MINIMIZE, dtype=np.int64)
        print(name, 'frames solved', len(res['fullpose']), 'minimize calls', len(N_MINIMIZE),
              'err keys', list(dbg['stageii_errs'].keys()))
    np.savez_compressed(os.path.join(HERE, 'ref_stageii.npz'), **out)
    print('wrote', os.path.join(HERE, 'ref_stageii.npz'))


if __name__ == '__main__':
    main()
