"""TEST INFRASTRUCTURE -- the full-length ORACLE trajectories (+ sensitivity envelopes) of BASELINE configs 3, 4 and 5, the companions of
make_oracle_trajectories.py (config 2's six bench seeds):

  config3_7000 / _7001   captures 7000 and 7001 of the config-3 subject (workload.make_face_job: SMPL-X, 89 markers incl. face / hand vertices, fingers +
                 jaw + 80 expression coefficients free = 194 unknowns per Step-2 solve; chmosh.py:560-567, 681-705), all 4000 frames
  mano_72 / _73  the two MANO hands of config 4 (34 / 33 markers, hand-PCA coefficients free, no pose prior), all 10 000 frames
  config5_1000   the first 8000 frames of config 5's 50 000-frame SMPL-H capture (seed 1000)

Per case: the NumPy oracle (oracle/stageii_oracle.py: stageii_chain, the restatement of chmosh.py:584-724) on the observations as they
are, and K = 3 further runs on observations perturbed by 1e-13 m; spread[f] = how far the perturbed runs end up from the first at
frame f (pose variables, translation and -- config 3 -- expression coefficients).  tests/parity_envelope.py turns that into the parity
criterion; the simulated markers of the oracle are recomputed from the stored pose by the tests (oracle forward), not stored.

States are stored rounded to 2^-36 (1.5e-11: four orders below the criterion's 1e-7) so that the files compress.

Run (CPU only):  python tests/golden/make_oracle_trajectories_configs.py [config3_7000 mano_72 mano_73 config5_1000]
  config 3: ~6 min per run; MANO: ~2 min; config 5: ~4 min + the 50 000-frame generator."""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
K = 3
EPS = 1e-13
CASES = ('config3_7000', 'config3_7001', 'mano_72', 'mano_73', 'config5_1000')


def case_inputs(name):
    """(m, prior, closest, coef, obs, vis, model_type, chain keyword arguments) of a case -- also imported by the GPU tests, which feed
    the same observations to the device."""
    from moshpp_amd import workload
    from oracle import stageii_oracle as so
    if name.startswith('config3_'):
        from tests.helpers import face_capture_host, face_job_oracle
        job = workload.make_face_job()
        m, pr, closest, coef = face_job_oracle(job)
        cap = face_capture_host(job, m, closest, coef, int(name.split('_')[1]), 4000)
        return dict(job=job, m=m, prior=pr, closest=closest, coef=coef, obs=cap['obs'], vis=cap['vis'], model_type='smplx',
                    kw=dict(optimize_fingers=True, optimize_face=True, free_shape='expr'))
    if name.startswith('mano_'):
        seed = int(name.split('_')[1])
        job = workload.make_job('mano', 10000, {72: 34, 73: 33}[seed], seed=seed, optimize_fingers=True)
        n = 10000
    elif name.startswith('config5_'):
        job = workload.make_job('smplh', 50000, 53, seed=int(name.split('_')[1]))
        n = 8000
    else:
        raise KeyError(name)
    sm = job['sm']
    m = so.prepare_model(dict(v_template=sm.v_template, shapedirs=sm.shapedirs, posedirs=sm.posedirs, weights=sm.weights,
                              J_regressor=sm.J_regressor, parents=sm.parents, body_dof=sm.body_dof, hand_dof=sm.hand_dof,
                              hands_mean=sm.hands_mean, selected_components=sm.selected_components), job['betas'])
    can = so.verts_forward(m, so.fullpose_from_pose(m, np.zeros(m['NP'])), np.zeros(3))
    closest, coef = so.transformed_coeffs(can, job['markers_latent'])
    return dict(job=job, m=m, prior=job['prior'], closest=closest, coef=coef, obs=job['obs'][:n], vis=job['vis'][:n],
                model_type=job['model_type'], kw=dict(optimize_fingers=job['optimize_fingers']))


def _run(args):
    name, k = args
    os.environ['OMP_NUM_THREADS'] = '1'
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(limits=1)
    except Exception:
        pass
    from oracle import stageii_oracle as so
    c = case_inputs(name)
    obs = c['obs']
    if k > 0:
        obs = obs + EPS * np.random.default_rng([977, k] + [ord(ch) for ch in name]).standard_normal(obs.shape)
    t0 = time.time()
    ref = so.stageii_chain(c['m'], c['prior'], c['closest'], c['coef'], obs, c['vis'], c['model_type'], **c['kw'])
    shape = np.asarray(ref['shape']) if 'shape' in ref and ref['shape'] is not None else None
    return name, k, np.asarray(ref['pose']), np.asarray(ref['trans']), shape, np.asarray(ref['iters']), np.asarray(ref['frame_ids']), time.time() - t0


def _q(x):
    return np.round(np.asarray(x, dtype=np.float64) * 2.0 ** 36) / 2.0 ** 36


def main():
    names = [a for a in sys.argv[1:] if not a.startswith('-')] or list(CASES)
    workers = int(os.environ.get('WORKERS', '4'))
    work = [(n, k) for n in names for k in range(K + 1)]
    res = {}
    with ProcessPoolExecutor(max_workers=min(workers, len(work))) as ex:
        for name, k, pose, trans, shape, iters, fids, dt in ex.map(_run, work):
            res[(name, k)] = (pose, trans, shape, iters, fids)
            print(f'{name} run {k}: {len(fids)} solved frames, {dt:.0f} s', flush=True)
            if all((name, kk) in res for kk in range(K + 1)):
                pose0, trans0, shape0, it0, fid0 = res[(name, 0)]
                spread = np.zeros(len(fid0))
                for kk in range(1, K + 1):
                    pose, trans, shape, it, fid = res[(name, kk)]
                    assert np.array_equal(fid, fid0)
                    spread = np.maximum(spread, np.abs(pose - pose0).max(1))
                    spread = np.maximum(spread, np.abs(trans - trans0).max(1))
                    if shape0 is not None and shape0.size:
                        spread = np.maximum(spread, np.abs(shape - shape0).max(1))
                fn = os.path.join(HERE, f'oracle_traj_{name}.npz')
                extra = {'shape': _q(shape0)} if shape0 is not None and shape0.size else {}
                np.savez_compressed(fn, pose=_q(pose0), trans=_q(trans0), iters=it0.astype(np.int16), frame_ids=fid0.astype(np.int32),
                                    spread=spread.astype(np.float32), eps=EPS, k=K, quantum=2.0 ** -36, **extra)
                print(f'{name}: spread <= 3e-9 on {(spread <= 3e-9).sum()} of {len(spread)} frames, max {spread.max():.2e}; '
                      f'{os.path.getsize(fn) / 1e6:.1f} MB', flush=True)


if __name__ == '__main__':
    main()
