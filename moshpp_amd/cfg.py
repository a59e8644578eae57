"""Configuration shim.

The reference hands `mosh_stageii` an OmegaConf `DictConfig` built from
support_data/conf/moshpp_conf.yaml (src/moshpp/mosh_head.py:543-559) and reads it with attribute AND item
access (chmosh.py:464-500, 535, 653; `cfg.moshpp[f'{cfg_key}']` at :477) and writes to it (:479, 485).
OmegaConf is not a dependency of this package: `mosh_stageii` accepts any object with that access pattern
(a real DictConfig included); `Cfg` / `make_cfg` provide one for standalone use, with the Stage-II
relevant defaults of the reference yaml (:13-47, 95-147).
"""
from __future__ import annotations

import copy

STAGEII_WEIGHTS = {   # moshpp_conf.yaml:118-125 (smplh == smplx), :166-173 (smplx_grab_vtemplate)
    'smplh': dict(stageii_wt_data=400, stageii_wt_velo=2.5, stageii_wt_dmpl=1.0, stageii_wt_expr=1.0,
                  stageii_wt_poseB=1.6, stageii_wt_poseH=1.0, stageii_wt_poseF=1.0, stageii_wt_annealing=2.5),
    'smplx': dict(stageii_wt_data=400, stageii_wt_velo=2.5, stageii_wt_dmpl=1.0, stageii_wt_expr=1.0,
                  stageii_wt_poseB=1.6, stageii_wt_poseH=1.0, stageii_wt_poseF=1.0, stageii_wt_annealing=2.5),
    'smplx_grab_vtemplate': dict(stageii_wt_data=400, stageii_wt_velo=2.5, stageii_wt_dmpl=1.0, stageii_wt_expr=0.9,
                                 stageii_wt_poseB=1.6, stageii_wt_poseH=0.4, stageii_wt_poseF=15.0,
                                 stageii_wt_annealing=2.5),
}


_STAGEI = dict(stagei_wt_poseH=3.0, stagei_wt_poseF=3.0, stagei_wt_expr=34.0, stagei_wt_pose=3.0, stagei_wt_poseB=3.0,
               stagei_wt_init_finger_left=400.0, stagei_wt_init_finger_right=400.0, stagei_wt_init_finger=400.0,
               stagei_wt_betas=10.0, stagei_wt_init=300, stagei_wt_data=75.0, stagei_wt_surf=10000.0,
               stagei_wt_annealing=[1.0, 0.5, 0.25, 0.125])          # moshpp_conf.yaml:104-117 (smplh == smplx)
STAGEII_WEIGHTS['smplh'].update(_STAGEI)
STAGEII_WEIGHTS['smplx'].update(_STAGEI)
STAGEII_WEIGHTS['smplx_grab_vtemplate'].update(dict(        # moshpp_conf.yaml:148-165
    stagei_wt_surf=10000.0, stagei_wt_init_hand=347.36, stagei_wt_init_finger=789.47, stagei_wt_init_finger_left=789.47,
    stagei_wt_init_finger_right=789.47, stagei_wt_init_head=220.69, stagei_wt_init_face=1100.0, stagei_wt_poseH=5.31,
    stagei_wt_poseF=28.97, stagei_wt_expr=6.99, stagei_wt_pose=3.0, stagei_wt_poseB=3.0, stagei_wt_betas=10.0, stagei_wt_init=300.0,
    stagei_wt_data=75.0, stagei_wt_annealing=[1.0, 0.5, 0.25, 0.125]))


class Cfg(dict):
    """dict with attribute access, recursively (enough of DictConfig for the Stage-II path)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_container(self):
        return {k: (v.to_container() if isinstance(v, Cfg) else v) for k, v in self.items()}

    def set_dotted(self, dotted, value):
        node = self
        parts = dotted.split('.')
        for p in parts[:-1]:
            if p not in node or not isinstance(node[p], Cfg):
                node[p] = Cfg()
            node = node[p]
        node[parts[-1]] = value


def default_cfg():
    """Stage-II relevant subset of support_data/conf/moshpp_conf.yaml with its default values."""
    return Cfg({
        'mocap': dict(fname=None, unit='mm', rotate=None, exclude_markers=None, only_markers=None, exclude_marker_types=None,
                      subject_id=-1, subject_name=None, multi_subject=False, start_fidx=0, end_fidx=-1, ds_rate=1),
        'surface_model': dict(type='smplx', fname=None, dmpl_fname=None, num_betas=16, betas_expr_start_id=300,
                              num_dmpls=8, dof_per_hand=24, num_expressions=80, use_hands_mean=True, gender='neutral'),
        'moshpp': dict(pose_body_prior_fname=None, pose_hand_prior_fname=None, optimize_fingers=False,
                       optimize_face=False, optimize_toes=False, optimize_betas=True, optimize_dynamics=False,
                       head_marker_corr_fname=None, betas_fname=None, v_template_fname=None, wrist_markers_on_stick=False,
                       separate_types=['body', 'face', 'finger'],
                       stagei_frame_picker=dict(type='random_strict', seed=100, num_frames=12, least_avail_markers=1.0,
                                                stagei_mocap_fnames=None),
                       verbosity=1, visualization=dict(marker_radius=dict(body=0.009, face=0.004, finger=0.005))),
        'dirs': dict(support_base_dir=None, work_base_dir=None, stagei_fname=None, stageii_fname=None, log_fname=None,
                     marker_layout=dict(fname=None)),
        'opt_settings': dict(weights_type=None, weights=None, maxiter=100, stagei_lr=1e-3, extra_initial_rigid_adjustment=False),
        # extensions of this implementation (absent in the reference; chain_mode 'auto': chmosh.StageIISolver.choose_chain_mode)
        'moshpp_amd': dict(chain_mode='auto', num_chunks=0, chunk_warmup=32, verify_tol=1e-9, device=None),
        'runtime': dict(stagei_only=False),
    })


def make_cfg(dict_cfg=None, **dotlist):
    """Merge order of MoSh.prepare_cfg (mosh_head.py:551-559): defaults <- dotted kwargs <- dict_cfg.
    `opt_settings.weights` resolves to the table named by `opt_settings.weights_type` (default: the
    surface-model type; yaml :96-97); only smplh / smplx / smplx_grab_vtemplate tables exist in the reference."""
    cfg = default_cfg()
    for k, v in dotlist.items():
        cfg.set_dotted(k, v)

    def merge(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), Cfg):
                merge(dst[k], v)
            else:
                dst[k] = v
    if dict_cfg:
        merge(cfg, dict_cfg)
    if cfg.opt_settings.weights is None:
        wt = cfg.opt_settings.weights_type or cfg.surface_model.type
        if wt not in STAGEII_WEIGHTS:
            raise KeyError(f"no opt_weights table for '{wt}' (the reference yaml has only {list(STAGEII_WEIGHTS)}); "
                           f"set opt_settings.weights_type or opt_settings.weights")
        cfg.opt_settings.weights = Cfg(STAGEII_WEIGHTS[wt])
    return cfg
