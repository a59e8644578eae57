"""Frame-by-frame deviation GPU vs oracle on the config-3 case (SMPL-X, 89 markers, fingers + jaw + 80 expressions)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from moshpp_amd import capi
from oracle import stageii_oracle as so
from tests.helpers import shape_case, device_case
H = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 21
brief = len(sys.argv) > 3
case = shape_case('smplx', F=H, M=89, E=80, seed=seed, kind='expr')
dev = device_case(case, optimize_fingers=True, optimize_face=True, shape_kind='expr')
o = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
t0 = time.perf_counter()
ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'], case['vis'], 'smplx',
                       optimize_fingers=True, optimize_face=True, free_shape='expr')
print('oracle seconds', time.perf_counter() - t0)
dp = np.abs(o['fullpose'] - ref['fullpose']).max(1)
ds = np.abs(o['shape'] - ref['shape']).max(1)
print('seed', seed, 'max dp', dp.max(), 'max ds', ds.max(), 'max data sse gpu', o['errs'][:, 0].max(), 'oracle', ref['errs']['data'].max(), 'iters equal', bool((o['iters'][:, 0] == ref['iters']).all()))
for t in range(0 if brief else H):
    print(t, f'{dp[t]:.2e} {ds[t]:.2e}', 'iters gpu', o['iters'][t, 0], 'oracle', ref['iters'][t], 'status', o['status'][t],
          'data sse gpu %.6e oracle %.6e' % (o['errs'][t, 0], ref['errs']['data'][t]))
