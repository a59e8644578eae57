"""Body-model loading and pose-variable layout, host side.

Mirror of `load_surface_model` (reference src/moshpp/models/smpl_fast_derivatives.py:52-166): reads the
SMPL-family model file, infers the model type from posedirs (:63-68), builds the hand-PCA map
`selected_components` / `hands_mean` (:80-128) and hands the arrays to libmoshii, where `SmplModelLBS`'s
forward and Jacobian (:169-263) live as HIP kernels.

Model pickles of the SMPL family contain chumpy objects; chumpy is not a dependency here, so the
unpickler below substitutes a stand-in that keeps only the numeric payload.
"""
from __future__ import annotations

import io
import os
import pickle
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

MODEL_TYPE_BY_NJOINT_PARMS = {69: 'smpl', 153: 'smplh', 162: 'smplx', 45: 'mano', 105: 'animal_horse',
                              102: 'animal_dog'}   # smpl_fast_derivatives.py:66-67
SUPPORTED_TYPES = ('smpl', 'smplh', 'smplx', 'mano')


class _ChStandIn:
    """Stand-in for pickled chumpy.Ch instances: exposes the stored array as `.r` / np.asarray()."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {'_state': state})

    @property
    def r(self):
        for key in ('x', '_x', 'a'):
            if key in self.__dict__:
                v = self.__dict__[key]
                return np.asarray(v.r if isinstance(v, _ChStandIn) else v)
        raise AttributeError('chumpy stand-in without numeric payload')

    def __array__(self, dtype=None, copy=None):
        a = self.r
        return a.astype(dtype) if dtype is not None else a


class _ModelUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split('.')[0] == 'chumpy':
            return _ChStandIn
        return super().find_class(module, name)


def _to_numpy(v):
    if isinstance(v, _ChStandIn):
        return v.r
    return v


def read_model_file(fname_or_dict):
    """-> dict of plain arrays (v_template, shapedirs, posedirs, weights, J_regressor, kintree_table, ...)."""
    if isinstance(fname_or_dict, dict):
        dd = dict(fname_or_dict)
    elif str(fname_or_dict).endswith('.npz'):
        with np.load(fname_or_dict, allow_pickle=True) as z:
            dd = {k: z[k] for k in z.files}
    else:
        assert str(fname_or_dict).endswith('.pkl'), ValueError('surface_model_fname could only be a pkl (or npz) file.')
        with open(fname_or_dict, 'rb') as f:
            dd = _ModelUnpickler(io.BytesIO(f.read()), encoding='latin-1').load()
    return {k: _to_numpy(v) for k, v in dd.items()}


def read_v_template(fname):
    """Vertices of a v_template mesh file (.ply ascii / .obj / .npy / .npz['v']) -- psbody.mesh stand-in."""
    if fname.endswith('.npy'):
        return np.load(fname)
    if fname.endswith('.npz'):
        return np.load(fname)['v']
    verts = []
    with open(fname, 'r', errors='ignore') as f:
        if fname.endswith('.obj'):
            for line in f:
                if line.startswith('v '):
                    verts.append([float(x) for x in line.split()[1:4]])
        else:
            n, in_header = 0, True
            for line in f:
                if in_header:
                    if line.startswith('format') and 'ascii' not in line:
                        raise ValueError('only ascii .ply v_template files are supported')
                    if line.startswith('element vertex'):
                        n = int(line.split()[-1])
                    if line.strip() == 'end_header':
                        in_header = False
                    continue
                if len(verts) < n:
                    verts.append([float(x) for x in line.split()[:3]])
    return np.asarray(verts, dtype=np.float64)


@dataclass
class SurfaceModel:
    """Arrays + pose-variable layout of one body model (what SmplModelLBS holds, :169-244)."""
    model_type: str
    v_template: np.ndarray
    shapedirs: np.ndarray
    posedirs: np.ndarray
    weights: np.ndarray
    J_regressor: np.ndarray
    parents: np.ndarray
    body_dof: int
    hand_dof: int = 0
    hands_mean: Optional[np.ndarray] = None
    selected_components: Optional[np.ndarray] = None
    f: Optional[np.ndarray] = None            # [T][3] surface triangles (Stage-I: vertex normals, point-to-surface distance)
    _device: object = field(default=None, repr=False)

    @property
    def V(self): return self.v_template.shape[0]

    @property
    def K(self): return len(self.parents)

    @property
    def NP(self): return self.body_dof + self.hand_dof

    @property
    def num_total_betas(self): return self.shapedirs.shape[-1]

    def fullpose(self, pose):
        """smpl_fast_derivatives.py:194-204 for pose[..., NP] -> fullpose[..., 3K]."""
        pose = np.asarray(pose, dtype=np.float64)
        if self.hand_dof == 0:
            return pose[..., :self.body_dof].copy()
        hand = self.hands_mean + pose[..., self.body_dof:self.body_dof + self.hand_dof].dot(self.selected_components)
        return np.concatenate([pose[..., :self.body_dof], hand], axis=-1)

    def device(self):
        """The libmoshii model handle (created on first use; raises without a GPU)."""
        if self._device is None:
            from . import capi
            self._device = capi.Model(self.v_template, self.shapedirs, self.posedirs, self.weights, self.J_regressor,
                                      self.parents, self.body_dof, self.hand_dof, self.hands_mean,
                                      self.selected_components)
        return self._device

    def new_device(self):
        """A libmoshii model handle of its own: betas and the free shape block are STATE of the handle (moshii_model_set_betas /
        set_free_shape), so every solver that sets them takes one for itself instead of sharing the cached `device()`."""
        from . import capi
        return capi.Model(self.v_template, self.shapedirs, self.posedirs, self.weights, self.J_regressor, self.parents,
                          self.body_dof, self.hand_dof, self.hands_mean, self.selected_components)


def load_surface_model(surface_model_fname, pose_hand_prior_fname=None, use_hands_mean=False, dof_per_hand=12,
                       v_template_fname=None, surface_model_type: str = None) -> SurfaceModel:
    """Same arguments and semantics as the reference's `load_surface_model` (:52-58)."""
    dd = read_model_file(surface_model_fname)
    posedirs = np.asarray(dd['posedirs'], dtype=np.float64)
    posedirs = posedirs.reshape(posedirs.shape[0], 3, -1)
    njoint_parms = posedirs.shape[2] // 3
    model_type = surface_model_type if surface_model_type else MODEL_TYPE_BY_NJOINT_PARMS[njoint_parms]
    if model_type not in SUPPORTED_TYPES:
        raise NotImplementedError(f'surface model type {model_type} is outside the MI355X Stage-II scope '
                                  f'(supported: {SUPPORTED_TYPES})')
    v_template = np.asarray(dd['v_template'], dtype=np.float64)
    if v_template_fname is not None:
        assert os.path.exists(v_template_fname), FileExistsError(v_template_fname)
        v_template = read_v_template(v_template_fname)
    kintree = np.asarray(dd['kintree_table']).astype(np.int64)
    K = kintree.shape[1]
    id_to_col = {int(kintree[1, i]): i for i in range(K)}
    parents = np.array([-1] + [id_to_col[int(kintree[0, i])] for i in range(1, K)], dtype=np.int32)
    Jreg = dd['J_regressor']
    if hasattr(Jreg, 'toarray'):
        Jreg = Jreg.toarray()
    Jreg = np.asarray(Jreg, dtype=np.float64)
    shapedirs = np.asarray(dd['shapedirs'], dtype=np.float64)
    weights = np.asarray(dd['weights'], dtype=np.float64)
    hands_mean = selected_components = None
    hand_dof = 0
    if model_type in ('smplx', 'smplh'):
        pose_body_dof = njoint_parms - 90 + 3                                   # :81
        assert pose_hand_prior_fname is not None, 'pose_hand_prior_fname is required for smplh / smplx'
        if isinstance(pose_hand_prior_fname, dict):
            hp = pose_hand_prior_fname
        else:
            assert pose_hand_prior_fname.endswith('.npz')
            with np.load(pose_hand_prior_fname) as z:
                hp = {k: z[k] for k in z.files}
        cl, cr = np.asarray(hp['componentsl']), np.asarray(hp['componentsr'])
        meanl = np.asarray(hp['hands_meanl']) if use_hands_mean else np.zeros(cl.shape[1])   # :88
        meanr = np.asarray(hp['hands_meanr']) if use_hands_mean else np.zeros(cr.shape[1])   # :92
        selected_components = np.vstack((np.hstack((cl[:dof_per_hand], np.zeros_like(cl[:dof_per_hand]))),
                                         np.hstack((np.zeros_like(cr[:dof_per_hand]), cr[:dof_per_hand]))))  # :95-97
        hands_mean = np.concatenate((meanl, meanr))
        hand_dof = 2 * dof_per_hand
    elif model_type == 'mano':
        pose_body_dof = 3
        comps = np.asarray(dd['hands_components'])
        hands_mean = np.zeros(comps.shape[1]) if use_hands_mean else np.asarray(dd['hands_mean'])   # :114 (sic)
        selected_components = np.vstack((comps[:dof_per_hand]))
        hand_dof = dof_per_hand
    else:
        pose_body_dof = njoint_parms + 3
    return SurfaceModel(model_type=model_type, v_template=v_template, shapedirs=shapedirs, posedirs=posedirs,
                        weights=weights, J_regressor=Jreg, parents=parents, body_dof=int(pose_body_dof),
                        hand_dof=int(hand_dof), hands_mean=hands_mean, selected_components=selected_components,
                        f=np.asarray(dd['f'], dtype=np.int64).reshape(-1, 3) if 'f' in dd and np.size(dd['f']) else None)
