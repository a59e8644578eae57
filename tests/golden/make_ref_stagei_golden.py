"""Generates tests/golden/ref_stagei.npz by EXECUTING the reference's own `mosh_stagei` and `prepare_mosh_markers_latent`
(/root/reference/src/moshpp/chmosh.py:83-455 and :57-80, function sources taken from the file, unmodified) on seeded synthetic
inputs, together with the reference code they drive:

  * marker_layout.edit_tools.marker_layout_load                              edit_tools.py:83-183
  * models.bodymodel_loader.load_moshpp_models (+ AliasedBetas)              bodymodel_loader.py:52-153
  * models.smpl_fast_derivatives.load_surface_model, SmplModelLBS.__init__   smpl_fast_derivatives.py:52-244
  * transformed_lm.TransformedCoeffs / TransformedLms                         transformed_lm.py:45-162
  * prior.gmm_prior_ch.create_gmm_body_prior / MaxMixtureComplete             gmm_prior_ch.py:42-134
  * rigid_transformations.perform_rigid_adjustment                            rigid_transformations.py:39-83
  * scan2mesh.mesh_distance_main.PtsToMesh / MeshDistanceSquared (compute_r, direction)      mesh_distance_main.py:158-297
  * scan2mesh.robustifiers.SignedSqrt, scan2mesh.ch_vert_normals (TriEdges, NormalizedNx3, TriNormals, VertNormals),
    scan2mesh.ch_cross_product.CrossProduct, scan2mesh.matlab

so that the Stage-I SCHEDULE -- which residual blocks exist in which annealing round and with which weights, which variables are
free when (root + body pose without the toes, fingers only in the last two rounds, shared betas, latent markers, translations),
how observed labels are matched to the layout per frame, the rigid initialisation, what is returned under which keys -- and the
surface-distance term (nearest part -> normal -> sign -> signed square root) are the reference's executed code, not a restatement.

What stands in for the absent third-party modules (and therefore stays "restated", oracle/stagei_oracle.py's header):
  * `chumpy`: the lazy stand-in of make_ref_stageii_golden.py (no automatic differentiation); `ch.minimize(method='dogleg')`: the
    oracle's minimize_dogleg on the residual vector the REFERENCE built, Jacobian by central differences of that residual;
  * `psbody.smpl.verts.verts_decorated(...)`.r: the oracle's LBS forward;
  * `psbody.mesh.Mesh.estimate_vertex_normals`: the oracle's vert_normals; `psbody.mesh.spatialsearch.aabbtree_nearest`: the oracle's
    exhaustive nearest_on_mesh (triangle, part code, point);
  * the native `sample2meshdist` (derivatives only) is never called: derivatives come from the differences above.

Run in the build container only (needs /root/reference); the npz is committed.  tests/test_ref_golden.py regenerates the inputs from
the seeds recorded here and holds the oracle (and, on a GPU, the kernels) to the recorded result.
"""
import ast
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.golden import make_ref_stageii_golden as H   # noqa: E402  (registers the chumpy / psbody.smpl / cv2 / loguru stand-ins)
from oracle import stagei_oracle as s1                  # noqa: E402

REF = H.REF


class Mesh(object):
    """psbody.mesh.Mesh stand-in: vertices, faces, estimate_vertex_normals (the oracle's: sum of the incident faces' scaled normals,
    normalised)."""
    def __init__(self, v=None, f=None, **_kw):
        self.v = np.asarray(v, dtype=np.float64)
        self.f = np.asarray(f)

    def estimate_vertex_normals(self):
        return s1.vert_normals(self.v, np.asarray(self.f, dtype=np.int64))


def _aabbtree_compute(v, f):
    return (np.asarray(v, dtype=np.float64), np.asarray(f, dtype=np.int64))


def _aabbtree_nearest(handle, pts):
    tri, part, near = s1.nearest_on_mesh(np.asarray(pts, dtype=np.float64), handle[0], handle[1])
    return tri.reshape(1, -1), part.reshape(1, -1), near


def _install():
    import scipy
    if not hasattr(scipy, 'array'):
        scipy.array = np.array          # mesh_distance_main.py:9 imports the alias SciPy removed
    sys.modules['psbody.mesh'].Mesh = Mesh
    H._module('psbody.mesh.spatialsearch', aabbtree_compute=_aabbtree_compute, aabbtree_nearest=_aabbtree_nearest)
    H._module('moshpp'); H._module('moshpp.models'); H._module('moshpp.prior'); H._module('moshpp.tools'); H._module('moshpp.marker_layout')
    H._module('moshpp.scan2mesh')
    H._module('moshpp.scan2mesh.mesh_distance', sample2meshdist=None)
    mods = {}
    mods['matlab'] = H.load_ref('moshpp.scan2mesh.matlab', 'scan2mesh/matlab.py')
    sys.modules['moshpp.scan2mesh'].matlab = mods['matlab']
    mods['robust'] = H.load_ref('moshpp.scan2mesh.robustifiers', 'scan2mesh/robustifiers.py')
    H.load_ref('moshpp.scan2mesh.ch_cross_product', 'scan2mesh/ch_cross_product.py')
    H.load_ref('moshpp.scan2mesh.ch_vert_normals', 'scan2mesh/ch_vert_normals.py')
    mods['mdm'] = H.load_ref('moshpp.scan2mesh.mesh_distance_main', 'scan2mesh/mesh_distance_main.py')
    H.load_ref('moshpp.models.smpl_fast_derivatives', 'models/smpl_fast_derivatives.py')
    H.load_ref('moshpp.prior.gmm_prior_ch', 'prior/gmm_prior_ch.py')
    mods['bml'] = H.load_ref('moshpp.models.bodymodel_loader', 'models/bodymodel_loader.py')
    mods['tlm'] = H.load_ref('moshpp.transformed_lm', 'transformed_lm.py')
    mods['rig'] = H.load_ref('moshpp.rigid_transformations', 'rigid_transformations.py')
    mods['lm'] = H.load_ref('moshpp.marker_layout.labels_map', 'marker_layout/labels_map.py')
    return mods


def _marker_layout_load(mods):
    """marker_layout_load (edit_tools.py:83-183): only this function's source is executed (the module imports torch / psbody)."""
    import os.path as osp_
    from collections import OrderedDict
    from pathlib import Path
    from typing import Dict, List, Union
    src = open(os.path.join(REF, 'marker_layout', 'edit_tools.py')).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'marker_layout_load'][0]

    class _Color:
        def __init__(self, *_): pass
        def range_to(self, other, n): return [_Color() for _ in range(n)]
        def get_rgb(self): return (0.0, 0.0, 0.0)
    ns = {'np': np, 'osp': osp_, 'json': json, 'OrderedDict': OrderedDict, 'Color': _Color, 'Markerlayout': dict, 'Dict': Dict,
          'List': List, 'Union': Union, 'Path': Path, 'general_labels_map': mods['lm'].general_labels_map,
          'logger': types.SimpleNamespace(info=lambda *a: None, debug=lambda *a: None)}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'edit_tools.py', 'exec'), ns)
    return ns['marker_layout_load']


def run_reference_stagei(model_type, n_verts, nb, n_markers, n_frames, seed, optimize_fingers=False, extra_rigid=False, head=0,
                         optimize_betas=True, betas_init=False, face=False, n_expr=0, expr_start=300):
    from pathlib import Path
    from typing import Dict, List, Union
    from sklearn.neighbors import NearestNeighbors
    from tests.golden.ref_inputs import stagei_case
    mods = _install()
    tmp = tempfile.mkdtemp(prefix='ref_stagei_')
    case = stagei_case(model_type, n_verts, nb, n_markers, n_frames, seed, tmp, finger_markers=optimize_fingers, head_markers=head,
                       betas_init=betas_init, face_markers=face)
    src = open(os.path.join(REF, 'chmosh.py')).read()
    fns = {n.name: n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef)}
    ns = {'np': np, 'ch': H.ch, 'logger': H._quiet, 'Path': Path, 'Union': Union, 'List': List, 'Dict': Dict, 'DictConfig': dict,
          'marker_layout_load': _marker_layout_load(mods), 'general_labels_map': mods['lm'].general_labels_map,
          'load_moshpp_models': mods['bml'].load_moshpp_models, 'TransformedCoeffs': mods['tlm'].TransformedCoeffs,
          'TransformedLms': mods['tlm'].TransformedLms, 'perform_rigid_adjustment': mods['rig'].perform_rigid_adjustment,
          'visualize_shape_estimate': None, 'flatten_list': lambda l: [x for s in l for x in s], 'NearestNeighbors': NearestNeighbors,
          'Mesh': Mesh, 'PtsToMesh': mods['mdm'].PtsToMesh}
    exec(compile(ast.Module(body=[fns['prepare_mosh_markers_latent'], fns['mosh_stagei']], type_ignores=[]), 'chmosh.py', 'exec'), ns)
    W = s1.stagei_weights_default()
    cfg = H.Cfg.of(dict(
        mocap=dict(exclude_markers=None, exclude_marker_types=None, only_markers=None),
        dirs=dict(marker_layout=dict(fname=case['layout_fname'])),
        moshpp=dict(optimize_betas=optimize_betas, optimize_fingers=optimize_fingers, optimize_face=face, optimize_toes=False, optimize_dynamics=False,
                    pose_hand_prior_fname=case['hand_prior_fname'], pose_body_prior_fname=case['body_prior_fname'], verbosity=0,
                    head_marker_corr_fname=case['head_corr_fname'], stagei_frame_picker=dict(num_frames=n_frames),
                    visualization=dict(marker_radius=dict(body=0.009))),
        surface_model=dict(fname=case['model_fname'], type=model_type, use_hands_mean=False, dof_per_hand=case['dof_per_hand'],
                           num_betas=nb, num_expressions=n_expr, betas_expr_start_id=expr_start),
        opt_settings=dict(maxiter=100, stagei_lr=1e-3, weights_type=model_type, extra_initial_rigid_adjustment=extra_rigid,
                          weights=dict(W))))
    del H.N_MINIMIZE[:]
    out = ns['mosh_stagei'](case['frames'], cfg, betas_fname=case['betas_fname'])
    return out, case, cfg


CASES = {   # name: (model type, vertices, free betas, markers, frames, seed, switches)
    'smplh_body': dict(mt='smplh', V=700, nb=4, M=16, F=3, seed=21),                       # BASELINE config 4's calibration part, small
    'smplh_extra_rigid': dict(mt='smplh', V=700, nb=3, M=14, F=2, seed=22, extra_rigid=True),   # opt_settings.extra_initial_rigid_adjustment (:230-232)
    'smplh_fingers': dict(mt='smplh', V=900, nb=3, M=22, F=2, seed=23, fingers=True),      # finger markers: poseH + finger ids in the last two rounds (:389-393)
    'smplh_head_corr': dict(mt='smplh', V=700, nb=3, M=16, F=2, seed=24, head=4),          # head_marker_corr_fname: init_<type> without 'head', init_head_corr (:252-266, 362-369)
    'smplh_fixed_betas': dict(mt='smplh', V=700, nb=3, M=14, F=2, seed=25, optimize_betas=False),   # optimize_betas off: no beta term, betas not free (:155, 374, 402)
    'smplh_betas_init': dict(mt='smplh', V=700, nb=3, M=14, F=2, seed=26, betas_init=True),         # betas_fname given, optimize_betas on: the solve starts from them (:93-98, 164-170)
    'smpl_body': dict(mt='smpl', V=700, nb=3, M=14, F=2, seed=27),                         # SMPL: pose_body_ids = all_pose_ids[3:], 69-d prior (:284-285)
    'mano_fingers': dict(mt='mano', V=500, nb=3, M=12, F=2, seed=28, fingers=True),        # MANO: no body ids / prior, pose_finger_ids = all_pose_ids[3:] (:306-307)
    # SMPL-X optimize_face: betas fixed (the reference refuses shared betas + per-frame expressions, :295-299), every frame's model with
    # its own betas vector, jaw ids 66:69 + the expression block free in the last two rounds, poseF / expr terms (:300-305, 394-398)
    'smplx_face': dict(mt='smplx', V=800, nb=4, M=18, F=2, seed=29, face=True, optimize_betas=False, E=5, expr_start=4),
}


def main():
    only = sys.argv[1:]
    fn_out = os.path.join(HERE, 'ref_stagei.npz')
    out = dict(np.load(fn_out, allow_pickle=False)) if (only and os.path.exists(fn_out)) else {}
    for name, cs in CASES.items():
        if only and name not in only:
            continue
        res, case, cfg = run_reference_stagei(cs['mt'], cs['V'], cs['nb'], cs['M'], cs['F'], cs['seed'],
                                              optimize_fingers=cs.get('fingers', False), extra_rigid=cs.get('extra_rigid', False),
                                              head=cs.get('head', 0), optimize_betas=cs.get('optimize_betas', True),
                                              betas_init=cs.get('betas_init', False), face=cs.get('face', False), n_expr=cs.get('E', 0),
                                              expr_start=cs.get('expr_start', 300))
        dbg = res['stagei_debug_details']
        out[f'{name}_args'] = np.array([cs['V'], cs['nb'], cs['M'], cs['F'], cs['seed'], int(cs.get('fingers', False)),
                                        int(cs.get('extra_rigid', False)), int(cs.get('head', 0)), int(cs.get('optimize_betas', True)),
                                        int(cs.get('betas_init', False)), int(cs.get('face', False)), int(cs.get('E', 0)),
                                        int(cs.get('expr_start', 300))], dtype=np.int64)
        out[f'{name}_model_type'] = np.array(cs['mt'])
        out[f'{name}_optimize_fingers_after'] = np.array(bool(cfg.moshpp.optimize_fingers))
        out[f'{name}_betas'] = np.asarray(res['betas'], dtype=np.float64)
        out[f'{name}_markers_latent'] = np.asarray(res['markers_latent'], dtype=np.float64)
        out[f'{name}_latent_labels'] = np.array(res['latent_labels'])
        out[f'{name}_markers_latent_vids'] = np.array([res['markers_latent_vids'][l] for l in res['latent_labels']], dtype=np.int64)
        out[f'{name}_keys'] = np.array(sorted(res.keys()))
        out[f'{name}_debug_keys'] = np.array(sorted(dbg.keys()))
        out[f'{name}_err_keys'] = np.array(list(dbg['stagei_errs'].keys()))
        out[f'{name}_errs'] = np.array([float(v) for v in dbg['stagei_errs'].values()])
        out[f'{name}_pose'] = np.array(dbg['opt_models_pose'], dtype=np.float64)
        out[f'{name}_trans'] = np.array(dbg['opt_models_trans'], dtype=np.float64)
        out[f'{name}_optimize_face_after'] = np.array(bool(cfg.moshpp.optimize_face))
        out[f'{name}_labels_obs'] = np.array(['|'.join(sorted(l)) for l in dbg['stagei_labels_obs']])
        out[f'{name}_minimize_calls'] = np.array(H.N_MINIMIZE, dtype=np.int64)
        print(name, 'minimize calls (n, rows, iterations, evaluations):', H.N_MINIMIZE, 'errs', {k: float(v) for k, v in dbg['stagei_errs'].items()},
              flush=True)
        np.savez_compressed(fn_out, **out)
    print('wrote', fn_out)


if __name__ == '__main__':
    main()
