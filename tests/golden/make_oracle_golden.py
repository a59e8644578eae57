"""Pins the oracle: runs oracle/stageii_oracle.py on small seeded cases and stores the results, so that
later edits of the oracle (or of the synthetic generators) cannot silently move the parity target.
NOTE the reference itself cannot run here (chumpy / psbody.smpl absent): these are oracle outputs, not
reference outputs -- parity with the reference stays 'unpinned' (DESIGN.md)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import stageii_oracle as so   # noqa: E402
from tests.helpers import oracle_case     # noqa: E402

CASES = {'smpl_41mk_10f': dict(model_type='smpl', F=10, M=41, seed=11, fingers=False),
         'smplh_53mk_8f': dict(model_type='smplh', F=8, M=53, seed=12, fingers=False),
         'mano_24mk_8f': dict(model_type='mano', F=8, M=24, seed=13, fingers=True)}

out = {}
for name, c in CASES.items():
    case = oracle_case(c['model_type'], F=c['F'], M=c['M'], seed=c['seed'], empty_frames=(3,))
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'],
                           c['model_type'], optimize_fingers=c['fingers'])
    out[f'{name}/fullpose'] = ref['fullpose']
    out[f'{name}/trans'] = ref['trans']
    out[f'{name}/iters'] = ref['iters']
    out[f'{name}/frame_ids'] = ref['frame_ids']
    out[f'{name}/err_data'] = ref['errs']['data']
    out[f'{name}/obs_checksum'] = np.array([case['obs'].sum(), case['vis'].sum(), case['coef'].sum()])
# Step-2 extras (chmosh.py:685-699): jaw + expression coefficients, DMPL coefficients
from tests.helpers import shape_case      # noqa: E402
SHAPE_CASES = {'smplx_expr5_6f': dict(model_type='smplx', kind='expr', E=5, F=6, M=40, seed=31),
               'smplh_dmpl4_6f': dict(model_type='smplh', kind='dmpl', E=4, F=6, M=40, seed=32)}
for name, c in SHAPE_CASES.items():
    case = shape_case(c['model_type'], F=c['F'], M=c['M'], E=c['E'], seed=c['seed'], kind=c['kind'])
    ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'],
                           c['model_type'], optimize_face=c['kind'] == 'expr', free_shape=c['kind'])
    out[f'{name}/fullpose'] = ref['fullpose']
    out[f'{name}/trans'] = ref['trans']
    out[f'{name}/shape'] = ref['shape']
    out[f'{name}/iters'] = ref['iters']
    out[f'{name}/err_shape'] = ref['errs']['shape']
    out[f'{name}/obs_checksum'] = np.array([case['obs'].sum(), case['vis'].sum(), case['coef'].sum()])
np.savez_compressed(os.path.join(HERE, 'oracle_golden.npz'), **out)
print({k: v.shape for k, v in out.items()})
