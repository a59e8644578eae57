"""Like for like: the oracle's Stage-I solve with its Jacobian taken by the same central differences as the executed-reference
fixture's stand-in for ch.minimize (stagei_solve(difference_jacobian=True)) against tests/golden/ref_stagei.npz -- iteration counts
of every solve, results, per-term SSE.  Minutes of CPU; its output is committed as tests/golden/ref_stagei_check.txt.
    python tests/golden/check_ref_stagei.py [case ...]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import stageii_oracle as so, stagei_oracle as s1                 # noqa: E402
from tests.test_ref_golden import STAGEI_REF_CASES, stagei_ref_case, oracle_errs_under_reference_keys   # noqa: E402

for name in (sys.argv[1:] or STAGEI_REF_CASES):
    for diff in (True, False):
        sc = stagei_ref_case(name, tempfile.mkdtemp())
        z = sc['ref']
        m = so.prepare_model(sc['pb']['model'])
        prior = so.prepare_gmm_prior(sc['pb']['gmm'], sc['npose']) if sc['npose'] else None
        st = {}
        got = s1.stagei_solve(m, sc['pb']['faces'], prior, sc['model_type'], sc['frames'], sc['vids'], sc['mask'], sc['m2b'], sc['nb'],
                              optimize_fingers=sc['fingers'], extra_initial_rigid_adjustment=sc['extra'], stats=st, difference_jacobian=diff,
                              head_corr=sc['head_corr'], betas_init=sc['betas_init'], optimize_face=sc['face'],
                              expr_start=sc['expr_start'] if sc['face'] else None, n_expr=sc['n_expr'])
        errs = oracle_errs_under_reference_keys(got['errs'], sc['mask'], drop_head=sc['head_corr'] is not None)
        nb = sc['nb']
        print(f"{name} [oracle Jacobian: {'central differences (as the fixture)' if diff else 'analytic'}]")
        print('  dogleg iterations per solve: oracle', st['per_call'], '| executed reference', z[f'{name}_minimize_calls'][:, 2].tolist())
        print(f"  max |difference|: betas {np.abs(got['betas'] - z[f'{name}_betas'][:nb]).max() if nb else 0.0:.2e}  markers_latent "
              f"{np.abs(got['markers_latent'] - z[f'{name}_markers_latent']).max():.2e} m  pose {np.abs(got['pose'] - z[f'{name}_pose']).max():.2e} rad  "
              f"trans {np.abs(got['trans'] - z[f'{name}_trans']).max():.2e} m  vids equal {np.array_equal(got['markers_latent_vids'], z[f'{name}_markers_latent_vids'])}")
        print('  SSE oracle   ', {k: float(f'{v:.6g}') for k, v in errs.items()})
        print('  SSE reference', {k: float(f'{v:.6g}') for k, v in zip(z[f'{name}_err_keys'].tolist(), z[f'{name}_errs'])}, flush=True)
