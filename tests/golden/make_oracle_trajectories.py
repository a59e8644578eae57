"""TEST INFRASTRUCTURE -- generates tests/golden/oracle_traj_seed<SEED>.npz: the ORACLE's Stage-II trajectory (oracle/stageii_oracle.py:
stageii_chain, the restatement of chmosh.py:584-724) over the WHOLE 4000-frame SMPL-H / 53-marker bench sequence of each of
bench.py's six seeds, plus the per-frame SENSITIVITY ENVELOPE the parity tests use instead of hand-listed frame windows:

  K further oracle runs on observations perturbed by 1e-13 m (Gaussian, far below anything a capture resolves) -> spread[f] =
  max over the runs of max_j |pose_k[f, j] - pose_0[f, j]|.  Where the chain is well conditioned the spread stays at round-off
  (< 1e-9 rad); where the reference's own algorithm sits on a knife edge (a dogleg step accepted / rejected on the last bit, a
  max-mixture component switch) it jumps by orders of magnitude for a stretch of frames and decays again.  A second float64
  implementation of the same formulas -- the HIP kernels -- is one more such perturbation: it must match the oracle to 1e-7 rad
  wherever spread <= 1e-9, and stay within ENVELOPE_FACTOR x the spread (and the north-star 1e-3 m marker RMSE) elsewhere.

Run (CPU only, ~2 minutes per oracle run, the runs of a seed in parallel):  python tests/golden/make_oracle_trajectories.py [seeds...]
Stored per seed (float32 where round-off is irrelevant to the stored quantity, ~2 MB): pose[F, NP] float64 -> float32 pair (hi, lo)
is avoided by storing float64 pose variables quantised to 2^-40 and compressed; spread[F] float32; iters[F] int16; frame_ids."""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
SEEDS = (1000, 123, 71, 5, 2024, 7)
K = 3                 # perturbed runs per seed
EPS = 1e-13           # metres
F, M = 4000, 53


def _run(args):
    seed, k = args
    os.environ['OMP_NUM_THREADS'] = '1'
    from moshpp_amd import workload
    from oracle import stageii_oracle as so
    job = workload.make_job('smplh', n_frames=F, n_markers=M, seed=seed)
    sm = job['sm']
    m = so.prepare_model(dict(v_template=sm.v_template, shapedirs=sm.shapedirs, posedirs=sm.posedirs, weights=sm.weights,
                              J_regressor=sm.J_regressor, parents=sm.parents, body_dof=sm.body_dof, hand_dof=sm.hand_dof,
                              hands_mean=sm.hands_mean, selected_components=sm.selected_components), job['betas'])
    pr = so.prepare_gmm_prior(job['seq']['gmm'], 63)
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3))
    closest, coef = so.transformed_coeffs(can, job['markers_latent'])
    obs = job['obs']
    if k > 0:
        obs = obs + EPS * np.random.default_rng(977 * seed + k).standard_normal(obs.shape)
    t0 = time.time()
    ref = so.stageii_chain(m, pr, closest, coef, obs, job['vis'], 'smplh')
    return seed, k, ref['pose'], ref['trans'], np.asarray(ref['iters']), np.asarray(ref['frame_ids']), time.time() - t0


def main():
    seeds = [int(a) for a in sys.argv[1:]] or list(SEEDS)
    work = [(s, k) for s in seeds for k in range(K + 1)]
    res = {}
    with ProcessPoolExecutor(max_workers=min(8, len(work))) as ex:
        for seed, k, pose, trans, iters, fids, dt in ex.map(_run, work):
            res[(seed, k)] = (pose, trans, iters, fids)
            print(f'seed {seed} run {k}: {len(fids)} solved frames, {dt:.0f} s', flush=True)
    for s in seeds:
        pose0, trans0, it0, fid0 = res[(s, 0)]
        spread = np.zeros(len(fid0))
        for k in range(1, K + 1):
            pose, trans, it, fid = res[(s, k)]
            assert np.array_equal(fid, fid0)
            spread = np.maximum(spread, np.abs(pose - pose0).max(1))
            spread = np.maximum(spread, np.abs(trans - trans0).max(1))
        fn = os.path.join(HERE, f'oracle_traj_seed{s}.npz')
        np.savez_compressed(fn, pose=pose0, trans=trans0, iters=it0.astype(np.int16), frame_ids=fid0.astype(np.int32),
                            spread=spread.astype(np.float32), eps=EPS, k=K)
        print(f'seed {s}: spread <= 1e-9 on {(spread <= 1e-9).sum()} of {len(spread)} frames, max {spread.max():.2e} rad; {os.path.getsize(fn) / 1e6:.1f} MB', flush=True)


if __name__ == '__main__':
    main()
