"""Oracle experiment (CPU): how far does a chunk that starts W frames early with the first-frame schedule
deviate from the exact sequential chain?  Basis for the chunked throughput mode's warm-up length."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from moshpp_amd import workload
from oracle import stageii_oracle as so

F = int(sys.argv[1]) if len(sys.argv) > 1 else 260
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
print('sequential', F, 'frames', time.time() - t0, 's', flush=True)
assert len(ref['frame_ids']) == F
res = {}
for s in (100, 180):
    for W in (0, 2, 4, 8, 16, 32, 64):
        a = s - W
        ch = so.stageii_chain(m, pr, closest, coef, job['obs'][a:a + W + 60], job['vis'][a:a + W + 60], 'smplh')
        d = np.abs(ch['fullpose'][W:] - ref['fullpose'][s:s + 60]).max(1)
        res[(s, W)] = d
        print(f's={s} W={W:3d} maxdiff frames 0..: ' + ' '.join(f'{x:.1e}' for x in d[[0, 1, 2, 4, 8, 16, 32, 59]]), flush=True)
np.savez('gpurun_out/chunk_dev.npz', **{f'{s}_{W}': v for (s, W), v in res.items()})
