"""One short sequential chain (for PMC collection): python tools/chain_small.py [F]"""
import sys
sys.path.insert(0, '.')
from moshpp_amd import workload
F = int(sys.argv[1]) if len(sys.argv) > 1 else 60
job = workload.make_job('smplh', F, 53, seed=1000)
solver = workload.make_solver(job)
out = solver.solve(job['obs'], job['vis'])
print('iters/frame', out['iters'][:, 0].mean())
