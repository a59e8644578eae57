"""CPU: the oracle's sequential chain over the full bench sequence (4000 frames) -> gpurun_out/oracle_full.npz"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from moshpp_amd import workload
from oracle import stageii_oracle as so
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
job = workload.make_job('smplh', n_frames=F, n_markers=53, seed=1000)
sm = job['sm']
m = so.prepare_model(dict(v_template=sm.v_template, shapedirs=sm.shapedirs, posedirs=sm.posedirs,
                          weights=sm.weights, J_regressor=sm.J_regressor, parents=sm.parents,
                          body_dof=sm.body_dof, hand_dof=sm.hand_dof, hands_mean=sm.hands_mean,
                          selected_components=sm.selected_components), job['betas'])
pr = so.prepare_gmm_prior(job['seq']['gmm'], 63)
can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3))
closest, coef = so.transformed_coeffs(can, job['markers_latent'])
t0 = time.time()
ref = so.stageii_chain(m, pr, closest, coef, job['obs'], job['vis'], 'smplh')
print('oracle', F, 'frames', time.time() - t0, 's')
np.savez('gpurun_out/oracle_full.npz', fullpose=ref['fullpose'], trans=ref['trans'], frame_ids=ref['frame_ids'], iters=ref['iters'],
         nvis=job['vis'].sum(1))
