"""Per-exchange wall-clock stamps of a cooperative chain (profile build: MOSHII_LIB=moshpp_amd/libmoshii_prof.so MOSHII_COOP=g).
For every exchange and rank: arrive, posted, everybody seen, payload read (100 MHz device clock).  Prints, per kind of exchange
(small = after an evaluation, large = normal equations): how long the ranks arrive apart, and what the protocol costs past the
last arrival."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
from moshpp_amd import capi, workload
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
G = int(os.environ.get('MOSHII_COOP', '0'))
if len(sys.argv) > 2 and sys.argv[2] == 'config3':   # the BASELINE config-3 subject (SMPL-X, 194 unknowns), one capture
    job = workload.make_face_job()
    solver = workload.make_solver(job)
    cap = workload.make_face_capture(job, solver, 7000, F)
    lib = capi.load()
    out = capi.chain_solve_host(solver.dev, solver.prior, solver.opts, [dict(attach=solver.attach, obs=cap['obs'], vis=cap['vis'], first=True)])[0]
    print(capi.last_launch_info())
else:
    job = workload.make_job('smplh', F, 53, seed=1000)
    solver = workload.make_solver(job)
    lib = capi.load()
    out = solver.solve(job['obs'], job['vis'])
N = 8192
buf = (C.c_longlong * (8 * N * 4))()
lib.moshii_prof_trace_read.argtypes = [C.POINTER(C.c_longlong)]
assert lib.moshii_prof_trace_read(buf) == N
t = np.array(buf, dtype=np.int64).reshape(8, N, 4)[:G]
# (an assembly's exchange in the reduce-scatter form spans two sequence numbers: its read-done stamp is the next one's)
for i in range(N - 1):
    if t[0, i, 0] > 0 and t[0, i, 3] == 0 and t[0, i + 1, 3] > 0:
        t[:, i, 3] = t[:, i + 1, 3]
used = (t[0, :, 0] > 0) & (t[0, :, 3] > 0)
idx = np.flatnonzero(used)[5:-5]
tt = t[:, idx, :].astype(np.float64) * 0.01     # us
arrive, posted, seen, done = tt[..., 0], tt[..., 1], tt[..., 2], tt[..., 3]
last_arrive = arrive.max(0)
skew = last_arrive - arrive.min(0)
after = done.max(0) - last_arrive                 # protocol + reads past the last arrival
dur0 = done[0] - arrive[0]
# classify: a large exchange takes longer to post (payload) -- use the sequence structure instead: gap to the previous exchange is not needed; post time works
post = (posted - arrive).mean(0)
large = post > np.median(post) * 1.5
for name, sel in (('small (after an evaluation)', ~large), ('large (normal equations)', large)):
    if sel.sum() == 0: continue
    print(f'{name}: n={sel.sum()}  ranks arrive {skew[sel].mean():.2f} us apart (max {skew[sel].max():.2f}); last arrival -> every rank done {after[sel].mean():.2f} us; '
          f'rank 0 spends {dur0[sel].mean():.2f} us in it; arrive->posted {post[sel].mean():.2f}, posted->seen (rank 0) {(seen[0]-posted[0])[sel].mean():.2f}, seen->read {(done[0]-seen[0])[sel].mean():.2f}')
    late = (arrive[:, sel] - arrive[:, sel].min(0)).mean(1)
    print('   mean lateness per rank (us):', np.round(late, 2))
print('frames', F, 'exchanges/frame', len(np.flatnonzero(used)) / F, 'us/frame', None)
