"""BASELINE config 4 end to end on one GPU: Stage-I on 12 picked frames (shape + latent markers), then Stage-II over a long capture of
the same subject with the Stage-I result (chunked mode).  Prints one JSON line.  Synthetic subject on the triangulated SMPL-H-sized body."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moshpp_amd import capi, synth                     # noqa: E402
from moshpp_amd.cfg import STAGEII_WEIGHTS              # noqa: E402
from moshpp_amd.chmosh import StageIISolver             # noqa: E402
from moshpp_amd.models import SurfaceModel              # noqa: E402
from moshpp_amd import workload                         # noqa: E402


def main(F2=50000):
    pb, dev, pr, kw = workload.make_stagei_job()
    mdl, M, nb = pb['model'], pb['M'], pb['nb']
    capi.stagei_solve_host(dev, pr, **kw)
    t = time.perf_counter(); s1 = capi.stagei_solve_host(dev, pr, **kw); t_s1 = time.perf_counter() - t
    # the long capture: the ground-truth subject (betas_gt, ml_gt) in smooth motion
    cl, coef = pb['attach_gt']
    pose_gt, trans_gt = synth.synth_motion(pb['NP'], mdl['body_dof'], F2, seed=3)
    pose_gt[:, mdl['body_dof']:] = 0; pose_gt[:, 30:36] = 0
    rng = np.random.default_rng(4)
    obs = np.zeros((F2, M, 3))
    t = time.perf_counter()
    for f in range(F2):
        vv = synth.lbs_numpy(pb['v_shaped_gt'], pb['J_gt'], mdl['posedirs'], mdl['weights'], mdl['parents'], pb['fullpose_of'](pose_gt[f]),
                             trans_gt[f], vids=cl.reshape(-1)).reshape(M, 3, 3)
        obs[f] = synth.markers_numpy(coef, vv[:, 0], vv[:, 1], vv[:, 2])
    obs += rng.normal(0, 0.0003, obs.shape)
    vis = rng.random((F2, M)) > 0.02
    t_gen = time.perf_counter() - t
    sm = SurfaceModel(model_type='smplh', v_template=mdl['v_template'], shapedirs=mdl['shapedirs'], posedirs=mdl['posedirs'],
                      weights=mdl['weights'], J_regressor=mdl['J_regressor'], parents=np.asarray(mdl['parents'], np.int32),
                      body_dof=mdl['body_dof'], hand_dof=mdl['hand_dof'], hands_mean=mdl['hands_mean'],
                      selected_components=mdl['selected_components'])
    betas = np.zeros(mdl['shapedirs'].shape[2]); betas[:nb] = s1['betas']
    from moshpp_amd.prior import create_gmm_body_prior
    prior = create_gmm_body_prior(pb['gmm'], exclude_hands=True)
    solver = StageIISolver(sm, betas, s1['markers_latent'], prior, dict(STAGEII_WEIGHTS['smplh']), surface_model_type='smplh',
                           num_betas=nb)
    solver.solve(obs[:2000], vis[:2000], chain_mode='chunked', verify_tol=1e-9)
    t = time.perf_counter(); out = solver.solve(obs, vis, chain_mode='chunked', verify_tol=1e-9); t_s2 = time.perf_counter() - t
    ok = out['status'] != 1
    rmse = float(np.sqrt((((out['markers_sim'] - obs) ** 2).sum(-1)[vis & ok[:, None]]).mean()))
    print(json.dumps({'workload': f'config 4 on one GPU: Stage-I (12 frames, {M} markers, {nb} betas) then Stage-II over {F2} frames',
                      'stagei_seconds': round(t_s1, 4), 'stagei_iterations': s1['iters'],
                      'stageii_seconds_incl_host_staging': round(t_s2, 3), 'stageii_frames_per_s': round(float(ok.sum()) / t_s2, 1),
                      'marker_rmse_to_observations_m': rmse, 'capture_generation_seconds_cpu': round(t_gen, 1)}))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 50000)
