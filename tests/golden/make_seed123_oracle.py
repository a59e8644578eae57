"""Fixture: the NumPy oracle's sequential chain on the 4000-frame bench sequence with seed 123 (SMPL-H, 53 markers), frames 0 .. 2199
(101 s of CPU) -- the sequence with the ill-conditioned stretch around frame 2160.  Stored: every 50th frame
of the whole run plus frames 2100 .. 2199 in full (fullpose, dogleg iteration counts).

    python tests/golden/make_seed123_oracle.py [--from tools/_oracle_seed123.npz]   ->  tests/golden/oracle_seed123.npz
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
F = 2200

if '--from' in sys.argv:      # re-slice an existing full trajectory (same generator: tools/dump_seq.py / this script)
    d = np.load(sys.argv[sys.argv.index('--from') + 1])
    fullpose, iters = d['fullpose'][:F], d['iters'][:F]
else:
    from moshpp_amd import workload
    from oracle import stageii_oracle as so
    import bench
    job = workload.make_job('smplh', n_frames=4000, n_markers=53, seed=123)   # the bench sequence (its noise draws depend on the length)
    m, pr, closest, coef = bench.oracle_setup(job)
    t0 = time.time()
    ref = so.stageii_chain(m, pr, closest, coef, job['obs'][:F], job['vis'][:F], 'smplh')
    print('oracle chain', F, 'frames:', time.time() - t0, 's')
    assert len(ref['frame_ids']) == F
    fullpose, iters = ref['fullpose'], ref['iters']
coarse = np.arange(0, F, 50)
np.savez_compressed(os.path.join(HERE, 'oracle_seed123.npz'), coarse_ids=coarse, coarse_fullpose=fullpose[coarse], coarse_iters=iters[coarse],
                    win_start=2100, win_fullpose=fullpose[2100:F], win_iters=iters[2100:F])
print('wrote oracle_seed123.npz')
