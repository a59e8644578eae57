"""Synthetic Stage-II jobs shaped like the BASELINE.json configs (bench.py, tests, smoke)."""
from __future__ import annotations

import numpy as np

from . import synth
from .cfg import STAGEII_WEIGHTS
from .models import load_surface_model
from .prior import create_gmm_body_prior

CONFIGS = {   # BASELINE.json "configs" (free variables / residual rows per frame)
    'config0_smpl_120f_41mk': dict(model_type='smpl', n_frames=120, n_markers=41, optimize_fingers=False),
    'config1_smplh_4000f_53mk': dict(model_type='smplh', n_frames=4000, n_markers=53, optimize_fingers=False),
    'config2_smplx_4000f_89mk': dict(model_type='smplx', n_frames=4000, n_markers=89, optimize_fingers=True),
    'config3_mano_10000f_33mk': dict(model_type='mano', n_frames=10000, n_markers=33, optimize_fingers=True),
}


def surface_model_from_synth(seq):
    """SurfaceModel through the same loader path a model file takes (models.load_surface_model)."""
    dd = {k: v for k, v in seq['model'].items() if not k.startswith('_')}
    return load_surface_model(dd, pose_hand_prior_fname=seq['hand_prior'], use_hands_mean=seq['use_hands_mean'],
                              dof_per_hand=seq['dof_per_hand'], surface_model_type=seq['model_type'])


def make_job(model_type='smplh', n_frames=4000, n_markers=53, seed=0, optimize_fingers=False, dd=None, **kw):
    """Host-side description of one sequence: SurfaceModel, prepared prior, betas, latent markers,
    obs[F,M,3] / vis[F,M] already ordered by latent label, Stage-II weights."""
    body_only = not optimize_fingers
    seq = synth.make_sequence(model_type, n_frames, n_markers, seed=seed, body_only_markers=body_only, dd=dd, **kw)
    sm = surface_model_from_synth(seq)
    prior = None
    if model_type != 'mano':
        prior = create_gmm_body_prior(seq['gmm'], exclude_hands=model_type in ('smplh', 'smplx'))
    obs = np.nan_to_num(seq['markers'])
    vis = ~np.isnan(seq['markers']).any(-1)
    return dict(seq=seq, sm=sm, prior=prior, betas=seq['betas'], markers_latent=seq['markers_latent'],
                obs=obs, vis=vis, weights=dict(STAGEII_WEIGHTS['smplh']), model_type=model_type,
                optimize_fingers=optimize_fingers)


def make_capture(job, solver, motion_seed, noise=0.0005, dropout=0.02, n_gaps=2):
    """Another capture of the SAME subject as `job` (same model, betas, priors, marker placement): a new seeded motion pushed
    through the solver's own attachment on the device (moshii_attach_markers, f64), plus the generator's noise / dropout model.
    Returns a job dict sharing everything but obs / vis with `job` -- seconds for 32 x 4000 frames, where the NumPy generator
    (synth.make_sequence) takes minutes."""
    seq, sm = job['seq'], job['sm']
    F, M = job['vis'].shape
    pose_gt, trans_gt = synth.synth_motion(sm.NP, sm.body_dof, F, seed=motion_seed)
    if not job['optimize_fingers'] and job['model_type'] != 'mano':
        pose_gt[:, sm.body_dof:] = 0.0
    if job['model_type'] == 'smplx':
        pose_gt[:, 66:75] = 0.0
    markers = solver.attach.markers(pose_gt, trans_gt)
    rng = np.random.default_rng(motion_seed + 5)
    markers += rng.normal(0, noise, markers.shape)
    drop = rng.random((F, M)) < dropout
    for _ in range(n_gaps):
        mk = rng.integers(M)
        s0 = rng.integers(max(1, F - 10))
        drop[s0:s0 + rng.integers(10, 50), mk] = True
    drop[0, :] = False
    markers[drop] = 0.0
    out = dict(job)
    out.update(obs=markers, vis=~drop, pose_gt=pose_gt, trans_gt=trans_gt, motion_seed=motion_seed)
    return out


def make_face_job(seed=26, n_markers=89, num_expressions=80, expr_boost=6.0, expr_decay=1.0):
    """The subject of BASELINE config 3: SMPL-X, 89 markers incl. face / hand vertices, fingers + jaw + `num_expressions` expression
    coefficients free in Step 2 (chmosh.py:560-567, 681-689) -- 194 unknowns per solve at the yaml default of 80.  The synthetic model
    carries the expression directions as shapedirs columns [16, 16 + E) (betas_expr_start_id = 16), boosted to centimetre scale so that
    the block is observable through the face markers.  Captures: make_face_capture."""
    E = int(num_expressions)
    dd = dict(synth.synth_model('smplx', seed=seed, num_betas=16 + E))
    sd = np.array(dd['shapedirs'], dtype=np.float64)
    sd[:, :, 16:] *= expr_boost / np.maximum(np.abs(sd[:, :, 16:]).max(axis=(0, 1), keepdims=True) / 0.005, 1e-12)
    sd[:, :, 16:] *= expr_decay ** np.arange(E)   # (a principal-component basis: the later directions move the surface less and less)
    dd['shapedirs'] = sd
    job = make_job('smplx', n_frames=4, n_markers=n_markers, seed=seed, optimize_fingers=True, dd=dd, num_betas=16 + E)
    job['betas'] = job['betas'].copy()
    job['betas'][16:] = 0.0                       # the subject's shape: the expression block belongs to the frames
    job.update(optimize_face=True, num_expressions=E, betas_expr_start_id=16)
    return job


def make_face_capture(job, solver, motion_seed, n_frames=4000, expr_amp=0.3, noise=0.0005, dropout=0.02, expr_vary=0.0):
    """One capture of the config-3 subject: seeded body + finger motion, a jaw motion and a per-capture expression (a constant offset
    of the free block: every frame has to find it, warm-started from its predecessor), generated on the device through a second
    model handle that carries the expression in its betas; the generator's noise / dropout model on top.
    expr_amp: round 3 used 0.6 -- on 2 of 5 captures the chain then lost track for a stretch (data SSE in the thousands); at 0.3 two float64
    executions (one workgroup / eight) agree to 1e-9 over 400 frames on 4 of 5 arbitrary captures (tools/config3_fixture.py,
    profiles/r04_config3_fixture.txt); weaker or decaying expression directions make the block LESS determined, not better behaved."""
    from . import capi
    sm = job['sm']
    E = job['num_expressions']
    rng = np.random.default_rng(motion_seed + 5)
    pose_gt, trans_gt = synth.synth_motion(sm.NP, sm.body_dof, n_frames, seed=motion_seed)
    t = np.arange(n_frames)[:, None] / 30.0
    pose_gt[:, 66:69] = 0.15 * np.sin(2 * np.pi * 1.1 * t + np.array([0.0, 1.0, 2.0]))   # jaw
    pose_gt[:, 69:75] = 0.0                                                               # eyes: never free
    gen = sm.new_device()
    b = solver.betas.copy()
    b[16:16 + E] = expr_amp * rng.standard_normal(E)
    gen.set_betas(b)
    att = capi.Attachment(gen, solver.tc.closest, solver.tc.coef)
    markers = att.markers(pose_gt, trans_gt)
    expr_gt = b[16:16 + E]
    if expr_vary > 0.0:
        # a facial expression that MOVES: every coefficient a slow sinusoid around the capture's offset.  The markers are linear in the
        # coefficients at a fixed pose, so the capture is the offset's markers plus the (pose-dependent) effect of the moving part,
        # taken from two more passes of the generator (offset +/- one unit of the time profile's basis would need E passes; the moving
        # part is instead a rank-one profile: all coefficients share one time course scaled per coefficient)
        g = expr_vary * expr_amp * 0.5 * rng.standard_normal(E)
        prof = np.sin(2 * np.pi * 0.35 * t[:, 0] + rng.random() * 6.0)
        b2 = b.copy(); b2[16:16 + E] += g
        gen.set_betas(b2)
        att2 = capi.Attachment(gen, solver.tc.closest, solver.tc.coef)
        m2 = att2.markers(pose_gt, trans_gt)
        att2.close()
        markers = markers + prof[:, None, None] * (m2 - markers)
        expr_gt = b[None, 16:16 + E] + prof[:, None] * g[None]
    att.close(); gen.close()
    markers += rng.normal(0, noise, markers.shape)
    drop = rng.random(markers.shape[:2]) < dropout
    drop[0, :] = False
    markers[drop] = 0.0
    return dict(obs=markers, vis=~drop, pose_gt=pose_gt, trans_gt=trans_gt, expr_gt=expr_gt)


def make_solver(job, maxiter=100):
    from .chmosh import StageIISolver
    return StageIISolver(job['sm'], job['betas'], job['markers_latent'], job['prior'], job['weights'],
                         surface_model_type=job['model_type'], optimize_fingers=job['optimize_fingers'], maxiter=maxiter,
                         optimize_face=job.get('optimize_face', False), betas_expr_start_id=job.get('betas_expr_start_id', 300),
                         num_expressions=job.get('num_expressions', 80))


class DeviceSequence:
    """One sequence with obs/vis and every output resident in HBM (torch tensors own the memory; libmoshii sees
    raw device pointers).  Used by bench.py and tools/ to time the hot path without host staging."""

    def __init__(self, job, solver, device):
        import torch
        from . import capi
        sm = job['sm']
        F, M = job['vis'].shape
        self.F, self.M, self.solver, self.job = F, M, solver, job
        self.obs = torch.from_numpy(np.ascontiguousarray(job['obs'])).to(device)
        self.vis = torch.from_numpy(np.ascontiguousarray(job['vis'].astype(np.uint8))).to(device)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        self.pose = z((F, sm.NP), torch.float64); self.fullpose = z((F, 3 * sm.K), torch.float64)
        self.trans = z((F, 3), torch.float64); self.msim = z((F, M, 3), torch.float64)
        self.errs = z((F, capi.NERR), torch.float64); self.iters = z((F, 2), torch.int32); self.status = z((F,), torch.int32)
        self.cdesc = (capi.ChainDesc * 1)()
        self.sdesc = (capi.SequenceDesc * 1)()
        for d in (self.cdesc[0], self.sdesc[0]):
            d.attach = solver.attach.handle; d.F = F
            d.obs = self.obs.data_ptr(); d.vis = self.vis.data_ptr()
            d.pose = self.pose.data_ptr(); d.fullpose = self.fullpose.data_ptr(); d.trans = self.trans.data_ptr()
            d.markers_sim = self.msim.data_ptr(); d.errs = self.errs.data_ptr(); d.iters = self.iters.data_ptr()
            d.status = self.status.data_ptr()
        self.cdesc[0].first_frame_schedule = 1
        self.report = None

    def _handles(self):
        s = self.solver
        return s.dev.handle, (s.prior.handle if s.prior is not None else None), s.opts[0]

    def solve_sequential(self, stream, coop=0):
        """coop = g > 0: the chain as a cooperative chain of g workgroups (capi.coop_group)."""
        import ctypes as C
        from . import capi
        mh, ph, opts = self._handles()
        capi.check(capi.load().moshii_chain_solve(mh, ph, C.byref(opts), 1, self.cdesc, capi.BUFFERS_DEVICE | capi.coop_group(coop), C.c_void_p(stream)))

    def solve_chunked(self, stream, num_chunks=0, warmup=32, verify_tol=1e-11, coop=0):
        """coop = g > 0: the repair sweeps as cooperative chains of g workgroups, launched by the host's rounds."""
        import ctypes as C
        from . import capi
        mh, ph, opts = self._handles()
        co = capi.ChunkOpts(int(num_chunks), int(warmup), float(verify_tol))
        rep = capi.ChunkReport()
        capi.check(capi.load().moshii_sequence_solve(mh, ph, C.byref(opts), 1, self.sdesc, C.byref(co), capi.BUFFERS_DEVICE | capi.coop_group(coop),
                                                     C.c_void_p(stream), C.byref(rep)))
        self.report = {k: getattr(rep, k) for k, _ in capi.ChunkReport._fields_}
        return self.report

    def results(self):
        return dict(pose=self.pose.cpu().numpy(), fullpose=self.fullpose.cpu().numpy(), trans=self.trans.cpu().numpy(),
                    markers_sim=self.msim.cpu().numpy(), errs=self.errs.cpu().numpy(), iters=self.iters.cpu().numpy(),
                    status=self.status.cpu().numpy())


def solve_many_chunked(seqs, stream, num_chunks=0, warmup=32, verify_tol=1e-11):
    """moshii_sequence_solve over several DeviceSequence objects of the same solver in ONE call (one launch for all
    their chunks).  Returns the chunk report."""
    import ctypes as C
    from . import capi
    mh, ph, opts = seqs[0]._handles()
    descs = (capi.SequenceDesc * len(seqs))()
    for i, sq in enumerate(seqs):
        C.memmove(C.byref(descs[i]), C.byref(sq.sdesc[0]), C.sizeof(capi.SequenceDesc))
    co = capi.ChunkOpts(int(num_chunks), int(warmup), float(verify_tol))
    rep = capi.ChunkReport()
    capi.check(capi.load().moshii_sequence_solve(mh, ph, C.byref(opts), len(seqs), descs, C.byref(co), capi.BUFFERS_DEVICE,
                                                 C.c_void_p(stream), C.byref(rep)))
    return {k: getattr(rep, k) for k, _ in capi.ChunkReport._fields_}


# ---- Stage-I -------------------------------------------------------------------------------------------------------
def make_stagei_job(model_type='smplh', n_verts=6890, nb=10, n_markers=53, n_frames=12, seed=1, dof_per_hand=24,
                    optimize_fingers=False):
    """Device handles + `capi.stagei_solve_host` keyword arguments of a seeded Stage-I problem (BASELINE config 4's calibration
    part: 12 picked frames, 53 markers, 10 betas on the triangulated SMPL-H-sized body).  Returns (problem, model, prior, kwargs)."""
    from . import capi
    from .cfg import STAGEII_WEIGHTS
    from .chmosh import stagei_pose_ids
    pb = synth.make_stagei_problem(model_type, n_verts=n_verts, nb=nb, M=n_markers, F=n_frames, seed=seed,
                                   dof_per_hand=dof_per_hand, finger_markers=optimize_fingers)
    mdl = pb['model']
    dev = capi.Model(mdl['v_template'], mdl['shapedirs'], mdl['posedirs'], mdl['weights'], mdl['J_regressor'], mdl['parents'],
                     mdl['body_dof'], mdl['hand_dof'], mdl['hands_mean'], mdl['selected_components'])
    prior = None
    if model_type != 'mano':
        g = create_gmm_body_prior(pb['gmm'], exclude_hands=model_type in ('smplh', 'smplx'))
        prior = capi.Prior(g['means'], g['chols'], g['weights'])
    W = STAGEII_WEIGHTS['smplh']
    pose_ids, body_ids, finger_ids = stagei_pose_ids(model_type, pb['NP'], optimize_fingers, False)
    M = pb['M']
    kw = dict(faces=pb['faces'], marker_vids=pb['vids'], m2b=np.ones(M) * pb['skin'], wt_init=np.ones(M) * W['stagei_wt_init'],
              frames=pb['frames'], nb=pb['nb'], weights=W, pose_ids=pose_ids, body_ids=body_ids if prior is not None else [],
              finger_ids=finger_ids)
    return pb, dev, prior, kw
