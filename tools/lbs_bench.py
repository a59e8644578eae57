"""Time the full-mesh LBS export kernels alone (for rocprofv3): python tools/lbs_bench.py [F] [reps] [model]"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from moshpp_amd import capi, workload
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mt = sys.argv[3] if len(sys.argv) > 3 else 'smplh'
M = {'smplh': 53, 'smpl': 41, 'smplx': 89, 'mano': 33}[mt]
# LBS_BODY=mesh: the synthetic body with the vertex order of a registered mesh (bone by bone, along each bone) instead of shuffled ids
import os
from moshpp_amd import synth
order = os.environ.get('LBS_BODY', 'shuffled')
job = workload.make_job(mt, 8, M, seed=1000, optimize_fingers=(mt == 'mano'), dd=synth.synth_model(mt, seed=1000, vertex_order=order))
solver = workload.make_solver(job)
sm = job['sm']
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
pose = torch.from_numpy(rng.normal(0, 0.3, (F, sm.NP)).astype(np.float32)).to(dev)
trans = torch.from_numpy(rng.normal(0, 1, (F, 3)).astype(np.float32)).to(dev)
verts = torch.empty((F, sm.V, 3), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
run = lambda: solver.dev.lbs_forward_device(F, pose.data_ptr(), trans.data_ptr(), verts.data_ptr(), C.c_void_p(stream))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3 / reps
out_bytes = F * sm.V * 12
print(f'{mt} [{order} vertex order] F={F}: {t*1e6:.1f} us per call, output {out_bytes/1e6:.1f} MB -> {out_bytes/t/1e9:.0f} GB/s ({out_bytes/t/8e12*100:.1f}% of 8 TB/s), {F/t:.0f} frames/s')

if os.environ.get('LBS_CHECK'):
    ref = solver.dev.lbs_forward(pose[:40].cpu().numpy().astype(np.float64), trans[:40].cpu().numpy().astype(np.float64))
    got = verts[:40].cpu().numpy()
    print(f'  check vs the f64 kernel on 40 frames: max |diff| {np.abs(got - ref).max():.2e} m')
if int(os.environ.get('MOSHII_LBS_STOP', '0')) & 16:
    torch.cuda.synchronize()
    raw = verts.view(-1)[:2 * 32 * 9].cpu().numpy().view(np.int64)
    st = raw[:256].reshape(8, 32)
    ps = raw[256:262]
    print('prep (workgroup 0, wave 0): ' + ' | '.join(f'{n} +{int(ps[k + 1] - ps[k])}' for k, n in enumerate(['hand PCA -> fullpose', 'Rodrigues + features', 'chain', 'transform rows out', 'feature pieces out'])))
    names = ['start', 'prologue done', 'k-loop done'] + [f'h{h} {w}' for h in range(8) for w in ('transforms in', 'blend done', 'exchange ready')] + ['rows out']
    for ti in range(7):
        row = st[ti]
        print(f'tile {ti}: ' + ' | '.join(f'{names[k]} +{int(row[k] - row[k - 1]) if k else 0}' for k in range(28)) + f' | [after k-loop: ring free +{int(row[28] - row[2])}, records in +{int(row[29] - row[28])}, transforms in +{int(row[3] - row[29])}] | total {int(row[27] - row[0])}' + (f' | gap to next {int(st[ti + 1][0] - row[27])}' if ti < 6 else ''))
