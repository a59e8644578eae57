import sys, numpy as np
sys.path.insert(0, '.')
from moshpp_amd import capi, workload
from tests.helpers import face_capture_host, face_job_oracle
F = int(sys.argv[1]) if len(sys.argv) > 1 else 400
H = 400
job = workload.make_face_job()
solver = workload.make_solver(job)
m, pr, closest, coef = face_job_oracle(job)
cap = face_capture_host(job, m, closest, coef, 7000, F)
g = np.load('tests/golden/config3_oracle.npz')
for NSEQ, coop in ((1, 0), (1, 1), (32, 0)):
    outs = capi.chain_solve_host(solver.dev, solver.prior, solver.opts,
                                 [dict(attach=solver.attach, obs=cap['obs'], vis=cap['vis'], first=True) for _ in range(NSEQ)], coop=coop)
    k = capi.last_launch_info()[0]
    same_copies = all(np.array_equal(o['fullpose'], outs[0]['fullpose']) for o in outs[1:])
    o = outs[0]
    same = o['iters'][:H, 0] == g['iters_7000'][:H]
    sel = np.arange(0, H, int(g['stride']))
    dpf = np.abs(o['fullpose'][sel] - g['fullpose_7000'][:len(sel)]).max(1)
    print(NSEQ, coop, k, 'copies identical', same_copies, 'first iters mismatch', int(np.argmax(~same)) if (~same).any() else -1, 'max dev', dpf.max(),
          'status', np.unique(o['status'], return_counts=True), 'first stride frame > 1e-7', int(sel[np.argmax(dpf > 1e-7)]) if (dpf > 1e-7).any() else -1)
