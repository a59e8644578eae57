"""MI355X-native MoSh++ Stage-II: drop-in for the reference's `chmosh.mosh_stageii`
(src/moshpp/chmosh.py:458-741).

`mosh_stageii` keeps the reference's name, signature, cfg fields, side effects on `cfg`
(:475-486) and return layout (:726-741), so it can be injected through the reference's own plugin
point:  `MoSh(**cfg).mosh_stageii(moshpp_amd.chmosh.mosh_stageii)`  (mosh_head.py:268-301).

Everything numeric runs in libmoshii (HIP, gfx950).  This file is host orchestration only: ingest,
model / prior / attachment setup, free-variable sets, chain construction and output assembly.
There is no CPU fallback: without the shared library or a GPU it raises.
"""
from __future__ import annotations

import logging

import numpy as np

from . import capi
from .mocap_interface import MocapSession, general_labels_map   # (the alias table of chmosh.py:466: moshpp_amd/data/label_aliases.json)
from .models import load_surface_model
from .prior import create_gmm_body_prior
from .transformed_lm import TransformedCoeffs

logger = logging.getLogger('moshpp_amd')

NUM_TRAIN_MARKERS = 46   # chmosh.py:460
ERR_KEYS = ('data', 'poseB', 'velo', 'poseH')


def _get(node, key, default=None):
    try:
        v = node[key]
    except Exception:
        return default
    return default if v is None else v


def read_dmpl_pcs(dmpl_fname):
    """`pickle.load(f)['eigvec']` of bodymodel_loader.load_dmpl / chmosh.py:511 -> [V, 3, n_dmpl]."""
    import pickle
    if isinstance(dmpl_fname, dict):
        return np.asarray(dmpl_fname['eigvec'], dtype=np.float64)
    if str(dmpl_fname).endswith('.npz'):
        with np.load(dmpl_fname) as z:
            return np.asarray(z['eigvec'], dtype=np.float64)
    with open(dmpl_fname, 'rb') as f:
        return np.asarray(pickle.load(f, encoding='latin-1')['eigvec'], dtype=np.float64)


def stageii_pose_ids(surface_model_type, pose_size, optimize_fingers, optimize_toes, optimize_face=False):
    """Free-variable index sets of chmosh.py:546-579 (sets), :645-647 / :665-667 (Step 1), :676-692 (Step 2)."""
    all_pose_ids = list(range(pose_size))
    pose_body_ids, pose_finger_ids, pose_face_ids = [], [], []
    pose_root_ids = all_pose_ids[:3]
    if surface_model_type == 'smpl':
        pose_body_ids = all_pose_ids[3:]
    elif surface_model_type == 'smplh':
        pose_body_ids = all_pose_ids[3:66]
        if optimize_fingers:
            pose_finger_ids = all_pose_ids[66:]
    elif surface_model_type == 'smplx':   # orient:3, body:63, jaw:3, eyel:3, eyer:3, handl, handr
        pose_body_ids = all_pose_ids[3:66]
        if optimize_face:
            pose_face_ids = all_pose_ids[66:69]             # the jaw (:562-563)
        if optimize_fingers:
            pose_finger_ids = all_pose_ids[75:]
    elif surface_model_type == 'mano':
        pose_finger_ids = all_pose_ids[3:]
    else:
        raise NotImplementedError(f'surface model type {surface_model_type}')
    step1 = pose_root_ids + pose_body_ids
    if len(pose_body_ids) and not optimize_toes:
        step1 = sorted(set(step1).difference(set(all_pose_ids[30:36])))
    step2 = list(step1)
    if optimize_fingers:
        step2 += pose_finger_ids
    step2 += pose_face_ids                                  # :685-689
    step2 = sorted(set(step2))
    return dict(root=pose_root_ids, body=pose_body_ids, finger=pose_finger_ids if optimize_fingers else [],
                face=pose_face_ids, step1=list(step1), step2=step2)


class StageIISolver:
    """Device-resident setup of one subject + marker layout: model, betas, prior, attachment, options.
    Reusable across sequences of the same subject (the reference rebuilds all of this per call)."""

    def __init__(self, surface_model, betas, markers_latent, prior, weights, surface_model_type=None,
                 num_betas=None, optimize_fingers=False, optimize_toes=False, maxiter=100,
                 optimize_face=False, betas_expr_start_id=300, num_expressions=80,
                 optimize_dynamics=False, num_dmpls=8, dmpl_pcs=None):
        """optimize_face (SMPL-X): jaw + `num_expressions` expression coefficients betas[expr_start:] become Step-2 free
        variables (chmosh.py:562-567, 685-689).  optimize_dynamics (SMPL / SMPL-H): `dmpl_pcs[V,3,>=num_dmpls]` replaces
        shapedirs[:, :, num_betas:num_betas+num_dmpls] and those coefficients become free (:507-514, 693-699)."""
        self.model_type = surface_model_type or surface_model.model_type
        self.shape_kind, self.shape_start, self.n_shape = None, 0, 0
        if optimize_dynamics:
            assert self.model_type in ['smpl', 'smplh'], \
                NotImplementedError('DMPLs are currently only supported by smpl and smplh models')   # :508-509
            if optimize_face:
                raise ValueError('optimize_face and optimize_dynamics are mutually exclusive (smplx vs smpl/smplh)')
            nb0 = int(num_betas)
            sd = np.array(surface_model.shapedirs, dtype=np.float64)
            if sd.shape[2] < nb0 + num_dmpls:
                sd = np.concatenate([sd, np.zeros(sd.shape[:2] + (nb0 + num_dmpls - sd.shape[2],))], axis=2)
            sd[:, :, nb0:nb0 + num_dmpls] = np.asarray(dmpl_pcs, dtype=np.float64)[:, :, :num_dmpls]   # :512-513
            import dataclasses
            surface_model = dataclasses.replace(surface_model, shapedirs=sd, _device=None)
            self.shape_kind, self.shape_start, self.n_shape = 'dmpl', nb0, int(num_dmpls)
        elif optimize_face:
            if self.model_type != 'smplx':
                raise ValueError('optimize_face needs an smplx model (chmosh.py:560-567)')
            self.shape_kind, self.shape_start, self.n_shape = 'expr', int(betas_expr_start_id), int(num_expressions)
            if self.shape_start + self.n_shape > surface_model.num_total_betas:
                raise ValueError(f'expression block [{self.shape_start}, {self.shape_start + self.n_shape}) exceeds the '
                                 f'{surface_model.num_total_betas} shape coefficients of the model')
        self.sm = surface_model
        # limits of the chain kernel's LDS layout (include/moshii.h), checked here with the numbers at hand
        M = int(np.asarray(markers_latent).shape[0])
        if M > capi.MAX_MARKERS:
            raise ValueError(f'{M} latent markers: libmoshii solves layouts of at most {capi.MAX_MARKERS} markers per subject '
                             f'(the per-frame Jacobian tiles and marker tables of a chain live in the 160 KiB LDS of one CU)')
        if surface_model.K > capi.MAX_JOINTS:
            raise ValueError(f'{surface_model.K} joints: libmoshii models have at most {capi.MAX_JOINTS} (ancestor sets are 64-bit masks)')
        self.dev = surface_model.new_device()   # a handle of its own: betas / free shape block are state of the handle
        betas = np.asarray(betas, dtype=np.float64).ravel()
        nb = len(betas) if num_betas is None else int(num_betas)
        b = np.zeros(surface_model.num_total_betas)
        b[:nb] = betas[:nb]                               # chmosh.py:499-500
        self.betas = b
        self.dev.set_betas(b)
        self.dev.set_free_shape(self.shape_start, self.n_shape)
        can_body = self.dev.lbs_forward(np.zeros((1, surface_model.NP)), np.zeros((1, 3)))[0]   # can_model.r (:502)
        self.can_body = can_body
        self.tc = TransformedCoeffs(can_body, markers_latent)
        self.attach = capi.Attachment(self.dev, self.tc.closest, self.tc.coef)
        self.ids = stageii_pose_ids(self.model_type, surface_model.NP, optimize_fingers, optimize_toes, optimize_face)
        n_unknowns = 3 + max(len(self.ids['step1']), len(self.ids['step2']) + self.n_shape)
        cap = capi.MAX_UNKNOWNS_EXTENDED if (self.n_shape or len(self.ids['face'])) else capi.MAX_UNKNOWNS
        if n_unknowns > cap:
            raise ValueError(f'{n_unknowns} unknowns per solve (3 + free pose variables + free shape coefficients): libmoshii solves at most '
                             f'{cap} for this configuration ({capi.MAX_UNKNOWNS} without / {capi.MAX_UNKNOWNS_EXTENDED} with jaw / free shape block)')
        self.prior = None
        if len(self.ids['body']):
            if prior is None:
                raise ValueError('a body pose prior is required for this model type (chmosh.py:613-614)')
            if prior['npose'] != len(self.ids['body']):
                raise ValueError(f"prior npose {prior['npose']} != len(pose_body_ids) {len(self.ids['body'])}")
            self.prior = capi.Prior(prior['means'], prior['chols'], prior['weights'])
        self.opts = capi.make_opts(weights, self.ids['step1'], self.ids['step2'], self.ids['body'], self.ids['finger'],
                                   maxiter=maxiter, num_train_markers=NUM_TRAIN_MARKERS, face_ids=self.ids['face'],
                                   n_shape=self.n_shape, shape_kind=self.shape_kind)
        self.optimize_fingers = bool(optimize_fingers)
        self.optimize_face = bool(optimize_face)
        self.optimize_dynamics = bool(optimize_dynamics)

    AUTO_MIN_FRAMES = 128     # below ~100 frames the chunk starts' 32 warm-up frames cost what the chunks save (profiles/r05_auto_threshold.txt:
                              # 64 frames 0.88x, 128 frames 1.6-1.7x, 256 frames 2-3x, 4000 frames 5.5-12.6x the sequential cooperative chain)

    def choose_chain_mode(self, n_frames, requested='auto'):
        """'auto' -> 'chunked' for a body-only solve of >= AUTO_MIN_FRAMES frames, 'sequential' otherwise: with finger / face / DMPL /
        shape coefficients free the chain's state has long memory, every fresh chunk start misses and the first repair sweep walks the
        whole sequence -- correct, but no faster than the sequential chain.  Any other request is returned as it is."""
        if requested != 'auto':
            return requested
        long_memory = self.optimize_fingers or self.optimize_face or self.optimize_dynamics or self.n_shape > 0
        return 'sequential' if (long_memory or n_frames < self.AUTO_MIN_FRAMES) else 'chunked'

    def solve(self, obs, vis, chain_mode='sequential', num_chunks=0, chunk_warmup=32, verify_tol=1e-9, init=None, coop_group=0):
        """obs[F,M,3], vis[F,M] -> per-frame arrays (rows of unsolved frames flagged by status != 0).
        chain_mode 'auto': choose_chain_mode(F) -- what mosh_stageii asks for by default; out['chain_mode'] says which one ran.
        chain_mode 'sequential': one chain, the reference's exact frame order (chmosh.py:584).
        chain_mode 'chunked': moshii_sequence_solve -- concurrent chunks with warm-up overlap, verified and
        repaired against the sequential chain to `verify_tol` (out['chunk_report']); free expression / DMPL coefficients travel in
        the hand-off states.  chain_mode 'chunked_host': the same scheme driven from the host (parallel.solve_sequence_chunked_host).
        coop_group: 0 = the library's choice (a lone chain / the repair sweeps of a chunked solve run as COOPERATIVE chains of several
        workgroups where that pays: body and finger solves), 1 = one workgroup per chain always, 2..8 = that many (capi.coop_group)."""
        F = obs.shape[0]
        # init = dict(pose, trans, pose_prev | None): continue a chain from that state (no first-frame schedule)
        ikw = {} if init is None else dict(init_pose=init['pose'], init_trans=init['trans'], init_pose_prev=init.get('pose_prev'))
        chain_mode = self.choose_chain_mode(F, chain_mode)
        if chain_mode == 'sequential' or F == 0:
            out = capi.chain_solve_host(self.dev, self.prior, self.opts,
                                        [dict(attach=self.attach, obs=obs, vis=vis, first=init is None, **ikw)], coop=coop_group)[0]
            out['chain_mode'] = 'sequential'
            return out
        if chain_mode == 'chunked':
            outs, report = capi.sequence_solve_host(self.dev, self.prior, self.opts, [dict(attach=self.attach, obs=obs, vis=vis, **ikw)],
                                                    num_chunks=num_chunks, warmup=chunk_warmup, verify_tol=verify_tol, coop=coop_group)
            outs[0]['chunk_report'] = report
            outs[0]['chain_mode'] = 'chunked'
            return outs[0]
        if chain_mode == 'chunked_host':
            # the chunk scheme driven from the host over ONE batched moshii_chain_solve per round (parallel.solve_sequence_chunked_host):
            # also carries the free shape coefficients across the hand-offs, which moshii_sequence_solve does not
            from .parallel import solve_sequence_chunked_host
            if init is not None:
                raise NotImplementedError("chain_mode='chunked_host' starts at the first frame")

            def solve_ranges(items):
                chains = []
                for a, b, st in items:
                    ch = dict(attach=self.attach, obs=obs[a:b], vis=vis[a:b], first=st is None)
                    if st is not None:
                        ch.update(init_pose=st['pose'], init_trans=st['trans'], init_pose_prev=st['pose_prev'])
                        if self.n_shape:
                            ch['init_shape'] = st['shape']
                    chains.append(ch)
                return capi.chain_solve_host(self.dev, self.prior, self.opts, chains)
            keys = ('pose', 'trans', 'shape') if self.n_shape else ('pose', 'trans')
            out, info = solve_sequence_chunked_host(solve_ranges, F, num_chunks or capi.device_cu_count(), warmup=chunk_warmup, verify_tol=verify_tol,
                                                    state_keys=keys)
            out['chunk_report'] = info
            out['chain_mode'] = 'chunked_host'
            return out
        raise ValueError(f'unknown chain_mode {chain_mode}')


def mosh_stageii(mocap_fname: str, cfg, markers_latent: np.ndarray, latent_labels: list, betas: np.ndarray,
                 marker_meta: dict, v_template_fname=None) -> dict:
    capi.load()
    capi.require_device()

    # 1. observed markers (chmosh.py:463-471)
    mocap = MocapSession(mocap_fname,
                         mocap_unit=cfg.mocap.unit,
                         mocap_rotate=cfg.mocap.rotate,
                         labels_map=general_labels_map,
                         only_subjects=[cfg.mocap.subject_name] if cfg.mocap.multi_subject else None)
    logger.debug('Loaded mocap markers for mosh stageii')

    # 2. switch finger / face optimisation off when layout or data cannot support it (:475-486)
    avail_labels = latent_labels
    for body_part, cfg_key in {'finger': 'optimize_fingers', 'face': 'optimize_face'}.items():
        if not cfg.moshpp[f'{cfg_key}']:
            continue
        if not np.any([body_part in m for m in marker_meta['marker_type_mask'].keys()]):
            cfg.moshpp[f'{cfg_key}'] = False
            logger.warning(f'{cfg_key} was activated but no {body_part} marker type detected in the marker layout: '
                           f'{cfg_key} = {cfg.moshpp[f"{cfg_key}"]}.')
        elif not np.any([(body_part in ltype) and l in avail_labels for l, ltype in marker_meta['marker_type'].items()]):
            cfg.moshpp[f'{cfg_key}'] = False
            logger.warning(f'{cfg_key} was activated but no {body_part} marker type detected in the mocaps: '
                           f'{cfg_key} = {cfg.moshpp[f"{cfg_key}"]}.')

    # 3. model, prior, attachment (:488-503)
    sm = load_surface_model(surface_model_fname=cfg.surface_model.fname,
                            surface_model_type=cfg.surface_model.type,
                            pose_hand_prior_fname=cfg.moshpp.pose_hand_prior_fname,
                            use_hands_mean=cfg.surface_model.use_hands_mean,
                            dof_per_hand=cfg.surface_model.dof_per_hand,
                            v_template_fname=v_template_fname)
    assert sm.model_type == cfg.surface_model.type, ValueError(f'{sm.model_type} != {cfg.surface_model.type}')
    prior = None
    if cfg.moshpp.pose_body_prior_fname and sm.model_type != 'mano':
        prior = create_gmm_body_prior(cfg.moshpp.pose_body_prior_fname,
                                      exclude_hands=sm.model_type in ['smplh', 'smplx'])
    stageii_wts = cfg.opt_settings.weights
    ext = _get(cfg, 'moshpp_amd', {}) or {}
    dmpl_pcs = None
    if cfg.moshpp.optimize_dynamics:   # :507-514 (the reference opens the pickle in text mode, a python-2 leftover)
        assert cfg.surface_model.type in ['smpl', 'smplh'], \
            NotImplementedError('DMPLs are currently only supported by smpl and smplh models')
        dmpl_pcs = read_dmpl_pcs(cfg.surface_model.dmpl_fname)
    solver = StageIISolver(sm, betas, markers_latent, prior, stageii_wts, surface_model_type=cfg.surface_model.type,
                           num_betas=cfg.surface_model.num_betas, optimize_fingers=cfg.moshpp.optimize_fingers,
                           optimize_toes=cfg.moshpp.optimize_toes, maxiter=cfg.opt_settings.maxiter,
                           optimize_face=cfg.moshpp.optimize_face,
                           betas_expr_start_id=_get(cfg.surface_model, 'betas_expr_start_id', 300),
                           num_expressions=_get(cfg.surface_model, 'num_expressions', 80),
                           optimize_dynamics=cfg.moshpp.optimize_dynamics,
                           num_dmpls=_get(cfg.surface_model, 'num_dmpls', 8), dmpl_pcs=dmpl_pcs)
    logger.debug(f'#observed, #simulated markers: {len(mocap.labels)}, {len(markers_latent)}')

    # 4. frames (:539-540) and the per-frame visible-label selection (:582-594) as arrays
    selected_frames = range(cfg.mocap.start_fidx, len(mocap) if cfg.mocap.end_fidx == -1 else cfg.mocap.end_fidx,
                            cfg.mocap.ds_rate)
    logger.debug(f'Starting mosh stageii for {len(selected_frames)} frames.')
    obs, vis = mocap.markers_aslabeled_arrays(latent_labels, selected_frames)

    # 5. the frame loop (:584-724) on the GPU
    # chain mode.  DEFAULT: 'auto' (StageIISolver.choose_chain_mode): a body-only solve of >= 128 frames runs 'chunked' -- the same
    # chain cut into concurrently solved chunks whose hand-offs are verified to `verify_tol` and repaired until the stitched result is
    # the sequential chain's (10x faster on the bench sequence) --, everything else (fingers / face / dynamics
    # free, short captures) 'sequential': the reference's literal frame order as one (cooperative) chain.  Why the chunked mode may
    # be the default since round 5: against the oracle's full-length trajectories and their sensitivity envelope
    # (tests/parity_envelope.py) the two modes have the SAME standing on every frame of every bench sequence -- both <= 1e-9 rad on the
    # well-conditioned frames, both part from the oracle on the same knife-edge stretches the oracle's own perturbed runs part on
    # (test_both_modes_lie_inside_the_oracle_envelope_on_every_frame_of_every_bench_seed; bench.py parity_every_frame).  What remains
    # particular to it: which chain repairs which chunk depends on timing, so the last digits can differ between runs (bounded by verify_tol;
    # measured: <= 1.1e-13 rad over 12 runs x 6 sequences, half of them bit-identical throughout -- profiles/r05_chunked_run_to_run.txt);
    # cfg.moshpp_amd.chain_mode = 'sequential' is the run-to-run bit-reproducible choice.
    default_mode = 'auto'
    out = solver.solve(obs, vis, chain_mode=_get(ext, 'chain_mode', default_mode),
                       num_chunks=int(_get(ext, 'num_chunks', 0)), chunk_warmup=int(_get(ext, 'chunk_warmup', 32)),
                       verify_tol=float(_get(ext, 'verify_tol', 1e-9)), coop_group=int(_get(ext, 'coop_group', 0)))
    logger.debug(f"stageii chain mode: {out.get('chain_mode')}")
    for fi in np.flatnonzero(out['status'] == 1):
        logger.error(f'no available observed markers for frame {selected_frames[fi]}. skipping the frame.')
    if np.any(out['status'] < 0):
        logger.warning(f"{int(np.sum(out['status'] < 0))} frames hit a non-positive-definite normal matrix "
                       f"(Cauchy step used instead of Gauss-Newton)")
    solved = np.flatnonzero(out['status'] != 1)

    # 6. outputs (:712-741)
    labels_arr = np.asarray(latent_labels, dtype=object)
    perframe = dict(markers_sim=[], markers_obs=[], labels_obs=[])
    for fi in solved:
        v = vis[fi]
        perframe['markers_sim'].append(out['markers_sim'][fi][v].copy())
        perframe['markers_obs'].append(obs[fi][v].copy())
        perframe['labels_obs'].append(labels_arr[v].tolist())
    errs = {'data': out['errs'][solved, 0]}
    if len(solver.ids['body']):
        errs['poseB'] = out['errs'][solved, 1]
    if len(solved) > 2:
        errs['velo'] = out['errs'][solved[2:], 2]   # the velocity term exists from the third solved frame on
    if solver.optimize_fingers:
        errs['poseH'] = out['errs'][solved, 3]
    if solver.optimize_face:
        errs['poseF'] = out['errs'][solved, 4]
        errs['expr'] = out['errs'][solved, 5]
    if solver.optimize_dynamics:
        if len(solved) > 1:
            errs['extrap_dmpl'] = out['errs'][solved[1:], 6]   # exists from the second solved frame on (:695)
        errs['dmpl'] = out['errs'][solved, 5]
    stageii_debug_details = {
        'stageii_errs': {k: np.array(v) for k, v in errs.items()},
        'markers_sim': perframe['markers_sim'],
        'markers_obs': perframe['markers_obs'],
        'labels_obs': perframe['labels_obs'],
        'markers_orig': mocap.markers[selected_frames],
        'labels_orig': mocap.labels,
        'mocap_fname': mocap_fname,
        'mocap_frame_rate': mocap.frame_rate,
        'mocap_time_length': mocap.time_length(),
        'stageii_iters': out['iters'][solved],          # extra: dogleg iterations / residual evaluations
        'stageii_solved_frame_ids': np.asarray(selected_frames)[solved] if len(solved) else np.zeros(0, dtype=int),
    }
    stageii_data = {'fullpose': out['fullpose'][solved].copy(), 'trans': out['trans'][solved].copy(),
                    'stageii_debug_details': stageii_debug_details}
    if solver.n_shape:
        # per-frame opt_model.betas: the frozen Stage-I betas with the free block's values added in
        betas_t = np.tile(solver.betas, (len(solved), 1))
        betas_t[:, solver.shape_start:solver.shape_start + solver.n_shape] += out['shape'][solved]
        if solver.optimize_dynamics:   # :721-722  betas[num_betas:total_num_betas]
            stageii_data['dmpls'] = betas_t[:, solver.shape_start:solver.shape_start + solver.n_shape].copy()
        if solver.optimize_face:       # :723-724  betas[exp_start_id:] -- the whole tail, as the reference stores it
            stageii_data['expression'] = betas_t[:, solver.shape_start:].copy()
    return stageii_data


# ---------------------------------------------------------------------------------------------------------------------
# Stage-I
# ---------------------------------------------------------------------------------------------------------------------
SMPLX_EYEBALL_VIDS = np.arange(9383, 10475)   # support_data/smplx_eyeballs.npz: the attachment never uses them (transformed_lm.py:49-50)


def stagei_pose_ids(surface_model_type, pose_size, optimize_fingers, optimize_toes):
    """chmosh.py:281-310, 383-388: (pose ids free in every round, body ids the prior sees, finger ids of the last two rounds)."""
    allp = list(range(pose_size))
    body, finger = [], []
    root = allp[:3]
    if surface_model_type == 'smpl':
        body = allp[3:]
    elif surface_model_type == 'smplh':
        body = allp[3:66]
        finger = allp[66:] if optimize_fingers else []
    elif surface_model_type == 'smplx':
        body = allp[3:66]
        finger = allp[75:] if optimize_fingers else []
    elif surface_model_type == 'mano':
        finger = allp[3:]
    else:
        raise NotImplementedError(f'Stage-I for surface model type {surface_model_type}')
    pose_ids = root + body
    if len(body) and not optimize_toes:
        pose_ids = sorted(set(pose_ids).difference(set(allp[30:36])))
    return pose_ids, body, finger


def mosh_stagei(stagei_frames, cfg, betas_fname=None, v_template_fname=None) -> dict:
    """Drop-in for the reference's `mosh_stagei` (src/moshpp/chmosh.py:83-455): same arguments, cfg fields, side effects on `cfg`
    (:103-137) and returned dict (:432-455).  `stagei_frames`: list of `label -> xyz` dicts, one per picked frame
    (frame_picker.load_marker_sessions_*).  The joint solve runs in libmoshii (moshii_stagei_solve, HIP); no CPU fallback."""
    from .marker_layout import marker_layout_load
    betas = None
    if betas_fname is not None:
        logger.debug(f'loading pre-computed betas: {betas_fname}')
        assert str(betas_fname).endswith('.npz'), ValueError(f'invalid numpy betas_fname: {betas_fname}')
        betas = np.load(betas_fname)['betas']
    excl_types = _get(cfg.mocap, 'exclude_marker_types')
    if cfg.surface_model.type == 'smplx' and cfg.moshpp.optimize_betas and (excl_types is not None and 'face' in excl_types):
        logger.info('Setting moshpp.optimize_face to False')      # :103-120
        cfg.moshpp.optimize_face = False
    layout = cfg.dirs.marker_layout.fname
    logger.info(f'using marker_layout_fname: {layout}')
    marker_meta = marker_layout_load(layout, include_nan=True, exclude_markers=_get(cfg.mocap, 'exclude_markers'),
                                     exclude_marker_types=excl_types, only_markers=_get(cfg.mocap, 'only_markers'),
                                     labels_map=general_labels_map)
    avail_labels = list(set(k for l in stagei_frames for k in list(l.keys())))
    for body_part, cfg_key in {'finger': 'optimize_fingers', 'face': 'optimize_face'}.items():      # :128-139
        if not cfg.moshpp[cfg_key]:
            continue
        if not np.any([body_part in m for m in marker_meta['marker_type_mask'].keys()]):
            cfg.moshpp[cfg_key] = False
            logger.warning(f'{cfg_key} was activated but no {body_part} marker type detected in the marker layout: {cfg_key} = False.')
        elif not np.any([(body_part in ltype) and l in avail_labels for l, ltype in marker_meta['marker_type'].items()]):
            cfg.moshpp[cfg_key] = False
            logger.warning(f'{cfg_key} was activated but no {body_part} marker type detected in the mocaps: {cfg_key} = False.')
    if cfg.moshpp.optimize_face and cfg.moshpp.optimize_betas and cfg.surface_model.type == 'smplx':
        raise NotImplementedError(                                                                     # :295-299
            'MoSh requires in the shape stage a single (shared) beta across frames and different per-frame facial expressions. '
            'So if you want to optimize the face you need to provide the shape, or provide a v_template.')
    sm = load_surface_model(surface_model_fname=cfg.surface_model.fname, surface_model_type=cfg.surface_model.type,
                            pose_hand_prior_fname=cfg.moshpp.pose_hand_prior_fname, use_hands_mean=cfg.surface_model.use_hands_mean,
                            dof_per_hand=cfg.surface_model.dof_per_hand, v_template_fname=v_template_fname)
    assert marker_meta['surface_model_type'] == sm.model_type == cfg.surface_model.type, ValueError(
        f"marker layout surface_model_type doesnt match that of curent mosh session surface_model.type: "
        f"{marker_meta['surface_model_type']} == {sm.model_type} == {cfg.surface_model.type}")
    if sm.f is None:
        raise ValueError('the surface model file has no faces (`f`): Stage-I needs the triangulated surface')
    prior = None
    if cfg.moshpp.pose_body_prior_fname and sm.model_type != 'mano':
        prior = create_gmm_body_prior(cfg.moshpp.pose_body_prior_fname, exclude_hands=sm.model_type in ['smplh', 'smplx'])
    optimize_betas = bool(cfg.moshpp.optimize_betas)
    num_betas = int(cfg.surface_model.num_betas)
    nb = num_betas if optimize_betas else 0
    all_betas = np.zeros(sm.num_total_betas)
    if betas is not None:
        all_betas[:num_betas] = np.asarray(betas, dtype=np.float64)[:num_betas]       # :164-170
    # the solver works on v_template + shapedirs[:, :, :nb] . betas: betas that stay fixed are folded into the template
    v_template = sm.v_template if optimize_betas else sm.v_template + sm.shapedirs.dot(all_betas)
    dev = capi.Model(v_template, sm.shapedirs, sm.posedirs, sm.weights, sm.J_regressor, sm.parents, sm.body_dof, sm.hand_dof,
                     sm.hands_mean, sm.selected_components)
    latent_labels = list(marker_meta['marker_vids'].keys())
    M = len(latent_labels)
    logger.debug(f'Estimating for #latent markers: {M}')
    m2b = np.ones(M) * 0.0095                                                          # :62-64
    for mask_type, mask in marker_meta['marker_type_mask'].items():
        m2b[np.asarray(mask, dtype=bool)] = marker_meta['m2b_distance'][mask_type]
    W = cfg.opt_settings.weights
    wt_init = np.zeros(M)
    for k, mask in marker_meta['marker_type_mask'].items():                            # :327-328 (before annealing)
        wt_init[np.asarray(mask, dtype=bool)] = _get(W, f'stagei_wt_init_{k}', W['stagei_wt_init'])
    frames, markers_obs, labels_obs = [], [], []
    lab_idx = {l: i for i, l in enumerate(latent_labels)}
    for obs_frame in stagei_frames:                                                    # :199-213
        obs_labels = [k for k, v in obs_frame.items() if not np.any(np.isnan(v))]
        common = [l for l in latent_labels if l in set(obs_labels)]                   # (the reference's set order is arbitrary)
        obf = np.vstack([obs_frame[k] for k in common]) if common else np.zeros((0, 3))
        frames.append((np.array([lab_idx[k] for k in common], dtype=np.int32), obf))
        markers_obs.append(obf); labels_obs.append(common)
    logger.debug('Number of available markers in each stagei selected frames: {}'.format(
        ', '.join([f'(F{fi:02d}, {len(fr)})' for fi, fr in enumerate(markers_obs)])))
    extra_rigid = bool(_get(cfg.opt_settings, 'extra_initial_rigid_adjustment', False))   # :230-232, done by the solver before round 1
    head_corr = None
    hfn = _get(cfg.moshpp, 'head_marker_corr_fname')
    if hfn is not None:                                                                # :252-266
        head_meta = np.load(hfn)
        if all(m in marker_meta['marker_vids'] for m in head_meta['mrk_labels']):
            head_corr = (np.array([lab_idx[str(m)] for m in head_meta['mrk_labels']], dtype=np.int32), np.asarray(head_meta['corr']))
            logger.info('Successfully took into account the correlation of the head markers')
    if head_corr is not None and 'head' in marker_meta['marker_type_mask']:
        # with the head correlation term the reference builds init_* for every type EXCEPT 'head' (`if k != 'head'`, :364): the
        # whole type leaves the plain init terms, not only the markers listed in the correlation file
        wt_init[np.asarray(marker_meta['marker_type_mask']['head'], dtype=bool)] = 0.0
    # weight of init_head_corr: wt_init.get('body', stagei_wt_init) (:366-367) -- the body type's weight only if the layout has one
    wt_init_head = float(_get(W, 'stagei_wt_init_body', W['stagei_wt_init'])) if 'body' in marker_meta['marker_type_mask'] \
        else float(W['stagei_wt_init'])
    pose_ids, body_ids, finger_ids = stagei_pose_ids(sm.model_type, sm.NP, cfg.moshpp.optimize_fingers, cfg.moshpp.optimize_toes)
    face_kw = {}
    if cfg.moshpp.optimize_face and sm.model_type == 'smplx':       # jaw + per-frame expressions in the last two rounds (:300-305, 394-398)
        face_kw = dict(n_expr=int(cfg.surface_model.num_expressions), expr_start=int(cfg.surface_model.betas_expr_start_id),
                       face_ids=list(range(66, 69)))
    if prior is None:
        body_prior_ids = []
    else:
        body_prior_ids = body_ids
    weights = {k: (list(W[k]) if k == 'stagei_wt_annealing' else float(W[k])) for k in
               ('stagei_wt_data', 'stagei_wt_poseB', 'stagei_wt_poseH', 'stagei_wt_betas', 'stagei_wt_surf', 'stagei_wt_annealing',
                'stagei_wt_expr', 'stagei_wt_poseF', 'stagei_wt_init')}
    pr_dev = capi.Prior(prior['means'], prior['chols'], prior['weights']) if prior is not None else None
    out = capi.stagei_solve_host(dev, pr_dev, faces=sm.f,
                                 marker_vids=list(marker_meta['marker_vids'].values()), m2b=m2b, wt_init=wt_init, frames=frames,
                                 nb=nb, weights=weights, pose_ids=pose_ids, body_ids=body_prior_ids, finger_ids=finger_ids,
                                 exclude_vids=SMPLX_EYEBALL_VIDS if sm.V == 10475 else None,
                                 betas_init=all_betas[:nb] if nb else None, maxiter=int(cfg.opt_settings.maxiter),
                                 stagei_lr=float(cfg.opt_settings.stagei_lr), head_corr=head_corr,
                                 wt_init_head=wt_init_head, extra_initial_rigid_adjustment=extra_rigid, **face_kw)
    if nb:
        all_betas[:nb] = out['betas']
    # stagei_errs: the SSE of every entry of the last round's opt_objs, under its keys and in its insertion order (:350-398, 415):
    # data, poseB, init_<type> per marker type of the layout (without 'head' when the head correlation term is on), init_head_corr,
    # beta, surf, then poseH / poseF / expr of the detailed rounds
    oe = out['errs']
    errs = {'data': oe['data']}
    if len(body_ids) and prior is not None:
        errs['poseB'] = oe['poseB']
    init_sq = np.asarray(out.get('init_sq', np.zeros(len(latent_labels))), dtype=np.float64)
    for k, mask in marker_meta['marker_type_mask'].items():
        if head_corr is not None and k == 'head':
            continue
        errs[f'init_{k}'] = float(init_sq[np.asarray(mask, dtype=bool)].sum())
    if head_corr is not None:
        errs['init_head_corr'] = oe['init_head_corr']
    if nb and not face_kw:
        errs['beta'] = oe['beta']
    errs['surf'] = oe['surf']
    if finger_ids:
        errs['poseH'] = oe['poseH']
    if face_kw:
        errs['poseF'] = oe['poseF']
        errs['expr'] = oe['beta']          # the shape block held the expressions
    # markers_latent_all_vids (:424-430): nearest vertex of the LAST frame's posed body for every valid marker of that frame
    b_last = (all_betas if optimize_betas else np.zeros_like(all_betas)).copy()
    if face_kw:     # opt_models[-1].r carries the last frame's expression coefficients (:300-305)
        b_last[face_kw['expr_start']:face_kw['expr_start'] + face_kw['n_expr']] += out['expression'][-1]
    dev.set_betas(b_last)
    last_body = dev.lbs_forward(out['pose'][-1:], out['trans'][-1:])[0]
    last = stagei_frames[-1]
    keys = [k for k, v in last.items() if not np.any(np.isnan(v))]
    all_vids = {}
    if keys:
        locs = np.array([np.asarray(last[k], dtype=np.float64) for k in keys])
        nearest = np.argmin(((locs[:, None, :] - last_body[None]) ** 2).sum(-1), axis=1)
        all_vids = {k: int(v) for k, v in zip(keys, nearest)}
    sim_all = out['markers_sim']
    stagei_debug_details = {'opt_models_trans': [t for t in out['trans']], 'opt_models_pose': [p for p in out['pose']],
                            'stagei_errs': errs, 'markers_latent_all_vids': all_vids,
                            'stagei_markers_sim_all': [sim_all[f] for f in range(len(frames))],
                            'stagei_markers_sim': [sim_all[f][ids] for f, (ids, _) in enumerate(frames)],
                            'stagei_markers_obs': markers_obs, 'stagei_labels_obs': labels_obs,
                            'stagei_iters': out['iters']}
    if face_kw:
        stagei_debug_details['opt_models_expression'] = [e for e in out['expression']]
    stagei_data = {'betas': all_betas, 'markers_latent': out['markers_latent'], 'latent_labels': latent_labels,
                   'marker_meta': marker_meta,
                   'markers_latent_vids': {l: int(v) for l, v in zip(latent_labels, out['markers_latent_vids'])}}
    if v_template_fname is not None:
        stagei_data['v_template_fname'] = v_template_fname
        stagei_debug_details['v_template'] = sm.v_template
    stagei_data['stagei_debug_details'] = stagei_debug_details
    return stagei_data
