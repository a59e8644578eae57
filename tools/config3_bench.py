"""BASELINE config 3 shape on one GPU: N independent SMPL-X sequences, 89 markers incl. face/hand vertices, fingers +
jaw + E expression coefficients free (3 + 111 + E unknowns; E = 80 is the reference yaml default -> NBLK 13, extended
kernel).  One chain per sequence, all in one launch.  python tools/config3_bench.py [N=32] [F=200] [E=80] [reps=3]
Prints frames/s, per-frame latency, and parity of chain 0 against the oracle on the first `--oracle` frames."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from moshpp_amd import capi                                    # noqa: E402
from oracle import stageii_oracle as so                        # noqa: E402  (checker only)
from tests.helpers import shape_case, device_case              # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F = int(sys.argv[2]) if len(sys.argv) > 2 else 200
E = int(sys.argv[3]) if len(sys.argv) > 3 else 80
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
NORACLE = 6

case = shape_case('smplx', F=F, M=89, E=E, seed=21, kind='expr')
dev = device_case(case, optimize_fingers=True, optimize_face=True, shape_kind='expr')
chains = [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True) for _ in range(N)]
capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], chains[:1])          # warm-up (module load)
best = 1e30
for _ in range(reps):
    t0 = time.perf_counter()
    outs = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], chains)
    best = min(best, time.perf_counter() - t0)
kern, lds, _ = capi.last_launch_info()
o = outs[0]
print(f'{N} x {F} frames SMPL-X 89 mk, unknowns 3+111+{E}: {N * F / best:.0f} frames/s (host buffers incl. staging), '
      f'{best / F * 1e6:.0f} us per frame step, kernel {kern}, LDS {lds} B, dogleg iters/frame {o["iters"][:, 0].mean():.1f}')
assert all(np.array_equal(x['fullpose'], o['fullpose']) for x in outs[1:]), 'identical inputs must give identical chains'
ref = so.stageii_chain(case['m'], case['prior'], case['closest'], case['coef'], case['obs'][:NORACLE], case['vis'][:NORACLE],
                       'smplx', optimize_fingers=True, optimize_face=True, free_shape='expr')
print(f'parity vs oracle on {NORACLE} frames: max|dpose| {np.abs(o["fullpose"][:NORACLE] - ref["fullpose"]).max():.2e} rad, '
      f'max|dexpr| {np.abs(o["shape"][:NORACLE] - ref["shape"]).max():.2e}, iters equal '
      f'{np.array_equal(o["iters"][:NORACLE, 0], ref["iters"])}')
