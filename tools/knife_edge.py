"""Is a chunked-vs-sequential excursion a knife-edge of the reference's algorithm?  Continue the sequential chain from its own state at
frame t0 (must reproduce it bit for bit) and from that state perturbed by eps: python tools/knife_edge.py SEED T0 T1 [EPS]."""
import sys
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import capi, workload
seed, t0, t1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
eps = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-13
job = workload.make_job('smplh', 4000, 53, seed=seed)
solver = workload.make_solver(job)
seq = solver.solve(job['obs'][:t1], job['vis'][:t1])
rng = np.random.default_rng(0)
for tag, d in (('exact state', 0.0), (f'state + {eps:g}', eps)):
    o = capi.chain_solve_host(solver.dev, solver.prior, solver.opts,
                              [dict(attach=solver.attach, obs=job['obs'][t0:t1], vis=job['vis'][t0:t1], first=False,
                                    init_pose=seq['pose'][t0 - 1] + d * rng.standard_normal(seq['pose'].shape[1]),
                                    init_trans=seq['trans'][t0 - 1], init_pose_prev=seq['pose'][t0 - 2])])[0]
    dev = np.abs(o['fullpose'] - seq['fullpose'][t0:t1]).max(1)
    first = np.flatnonzero(dev > 1e-7)
    it_diff = np.flatnonzero(o['iters'][:, 0] != seq['iters'][t0:t1, 0])
    print(f'{tag}: max |dpose| {dev.max():.3e} rad; first frame > 1e-7: {t0 + first[0] if len(first) else None}; '
          f'first frame with another dogleg iteration count: {t0 + it_diff[0] if len(it_diff) else None}; '
          f'marker rmse of the excursion {np.sqrt(((o["markers_sim"] - seq["markers_sim"][t0:t1]) ** 2).sum(-1).mean()):.2e} m')
