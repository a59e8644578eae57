"""Experiment behind profiles/r06_outriders_experiment.txt: does a chunk chain seeded from a STALE TRUE state (the sequential chain's state
`lag` frames before the chunk's warm-up starts) land on the sequential trajectory, where the fresh-start chunk of pass 1 does not?
usage (GPU): python tools/respec_experiment.py [seed [warmup]]"""
import sys, os
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import workload, capi
capi.load()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 71
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
job = workload.make_job('smplh', 4000, 53, seed=seed)
s = workload.make_solver(job)
obs, vis = job['obs'], job['vis']
true = s.solve(obs, vis, chain_mode='sequential')
P, T = true['pose'], true['trans']
F = len(P)
C = 16
starts = list(range(C * 4, F - C, C))
def run(chains):
    return capi.chain_solve_host(s.dev, s.prior, s.opts, chains)
def dev_at(outs, metas):
    r = []
    for o, (a, b) in zip(outs, metas):
        r.append(max(np.abs(o['pose'][-1] - P[b - 1]).max(), np.abs(o['trans'][-1] - T[b - 1]).max()))
    return np.array(r)
# fresh starts (pass 1): chain over [st - W, st + C) with the first-frame schedule
metas = [(st - W, st + C) for st in starts]
fresh = dev_at(run([dict(attach=s.attach, obs=obs[a:b], vis=vis[a:b], first=True) for a, b in metas]), metas)
print('seed', seed, 'W', W, 'chunks', len(starts), 'fresh-start misses (>1e-9):', int((fresh > 1e-9).sum()), 'gross (>1e-6):', int((fresh > 1e-6).sum()), flush=True)
# which of the two local solutions is the better one, and where they differ
fo = run([dict(attach=s.attach, obs=obs[a:b], vis=vis[a:b], first=True) for a, b in metas])
g = fresh > 1e-6
ef = np.array([o['errs'][-1][:4].sum() for o in fo]); et = np.array([true['errs'][b - 1][:4].sum() for a, b in metas])
dp = np.array([np.abs(o['pose'][-1] - P[b - 1]) for o, (a, b) in zip(fo, metas)])[g]
top = np.argsort(-np.median(dp, 0))[:6]
print(f'   of the {int(g.sum())} gross misses the fresh solution has the lower objective (data + prior + velocity + hand) in {int((ef[g] < et[g]).sum())}, the higher in {int((ef[g] > et[g]).sum())}; '
      f'median objective {np.median(ef[g]):.4g} (fresh) vs {np.median(et[g]):.4g} (sequential); pose variables with the largest median difference: '
      f'{[(int(i), round(float(np.median(dp[:, i])), 3)) for i in top]}', flush=True)
for lag in (0, 16, 48, 96, 192, 384):
    ch, mt = [], []
    for st in starts:
        a, b = st - W, st + C
        t0 = a - 1 - lag          # the stale true state: that of frame t0 (lag 0 = the exact predecessor state -> the sequential chain itself)
        if t0 < 1: continue
        ch.append(dict(attach=s.attach, obs=obs[a:b], vis=vis[a:b], first=False, init_pose=P[t0], init_trans=T[t0], init_pose_prev=P[t0 - 1]))
        mt.append((a, b))
    d = dev_at(run(ch), mt)
    fr = fresh[len(starts) - len(mt):]
    gross_f = fr > 1e-6
    print(f'lag {lag:4d}: misses {int((d > 1e-9).sum()):3d} gross {int((d > 1e-6).sum()):3d} of {len(mt)} | of the {int(gross_f.sum())} chunks the fresh start misses grossly: '
          f'{int(((d <= 1e-9) & gross_f).sum())} now exact | of the {int((~gross_f).sum())} others: {int(((d > 1e-6) & ~gross_f).sum())} now gross', flush=True)
    # variant: stale pose, but the translation of a rigid fit is not available here: take the stale state's pose with the TRUE translation of a - 1 (upper bound for a "tracked translation")
    ch2 = []
    for (a, b) in mt:
        t0 = a - 1 - lag
        ch2.append(dict(attach=s.attach, obs=obs[a:b], vis=vis[a:b], first=False, init_pose=np.concatenate([P[a - 1][:3], P[t0][3:]]), init_trans=T[a - 1], init_pose_prev=None))
    d2 = dev_at(run(ch2), mt)
    print(f'          (root orientation + translation of the true predecessor, joints from the stale state) misses {int((d2 > 1e-9).sum()):3d} gross {int((d2 > 1e-6).sum()):3d} | '
          f'fresh-gross now exact {int(((d2 <= 1e-9) & gross_f).sum())} | others now gross {int(((d2 > 1e-6) & ~gross_f).sum())}', flush=True)
