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
if os.environ.get('LBS_V'):   # (experiment: another vertex count, e.g. 6912 = 108 x 64: 128-byte-aligned output rows)
    synth.MODEL_DIMS = dict(synth.MODEL_DIMS); synth.MODEL_DIMS[mt] = (int(os.environ['LBS_V']), synth.MODEL_DIMS[mt][1])
job = workload.make_job(mt, 8, M, seed=1000, optimize_fingers=(mt == 'mano'), dd=synth.synth_model(mt, seed=1000, vertex_order=order))
solver = workload.make_solver(job)
sm = job['sm']
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
pose_h = rng.normal(0, 0.3, (F, sm.NP)).astype(np.float32)
if os.environ.get('LBS_HANDS') == 'still':   # a body-only Stage-II result (the reference's default): the hand-pose variables are the same in every frame
    pose_h[:, sm.body_dof:] = 0.0
pose = torch.from_numpy(pose_h).to(dev)
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
print(f'{mt} [{order} vertex order{", still hands" if os.environ.get("LBS_HANDS") == "still" else ""}] F={F}: {t*1e6:.1f} us per call, output {out_bytes/1e6:.1f} MB -> {out_bytes/t/1e9:.0f} GB/s ({out_bytes/t/8e12*100:.1f}% of 8 TB/s), {F/t:.0f} frames/s')

if os.environ.get('LBS_CHECK'):
    ref = solver.dev.lbs_forward(pose[:40].cpu().numpy().astype(np.float64), trans[:40].cpu().numpy().astype(np.float64))
    got = verts[:40].cpu().numpy()
    print(f'  check vs the f64 kernel on 40 frames: max |diff| {np.abs(got - ref).max():.2e} m')
if int(os.environ.get('MOSHII_LBS_STOP', '0')) & 16:
    torch.cuda.synchronize()
    lib = capi.load()
    lib.moshii_internal_l32.restype = C.c_void_p
    lib.moshii_internal_l32.argtypes = [C.c_void_p]
    buf_ = (C.c_longlong * 1024)()
    lib.moshii_internal_lbs_debug_times.argtypes = [C.c_void_p, C.c_void_p]
    lib.moshii_internal_lbs_debug_times(lib.moshii_internal_l32(solver.dev.handle), buf_)
    raw = np.array(buf_[:], dtype=np.int64)
    st = raw[:192].reshape(8, 24)
    ps = raw[192:202]
    if ps[6]:
        print(f'prep first loads (cycles after start): joint record {int(ps[6] - ps[0])}, this frame\'s pose variables {int(ps[7] - ps[0])}, frame 0\'s {int(ps[8] - ps[0])}, component window {int(ps[9] - ps[0])}')
    print('prep (workgroup 0, wave 0): ' + ' | '.join(f'{n} +{int(ps[k + 1] - ps[k])}' for k, n in enumerate(['hand PCA -> fullpose', 'Rodrigues + features', 'chain', 'transform rows out', 'feature pieces out'])))
    names = ['start', 'tables + first loads issued', 'k-loop done'] + [f'block {h} done' for h in range(8)] + ['rows out']
    for ti in range(7):
        row = st[ti]
        if ti < 6 and st[ti + 1][11]:
            print(f'  (shader clock between the starts of tiles {ti} and {ti + 1}: {(st[ti + 1][0] - row[0]) / max(1, st[ti + 1][12] - row[12]) * 0.1:.2f} GHz)')
        if row[11] == 0:
            break
        print(f'tile {ti}: ' + ' | '.join(f'{names[k]} +{int(row[k] - row[k - 1]) if k else 0}' for k in range(12)) + f' | [block 3: wait + round-0 matrix instructions + next loads issued +{int(row[16] - row[5])}, previous rows read +{int(row[17] - row[16])}, stored +{int(row[13] - row[17])}, further rounds +{int(row[14] - row[13])}, apply + exchange +{int(row[15] - row[14])}, barrier +{int(row[6] - row[15])}]' + f' | total {int(row[11] - row[0])}' + (f' | gap to next {int(st[ti + 1][0] - row[11])}' if ti < 6 and st[ti + 1][11] else ''))
if int(os.environ.get('MOSHII_LBS_STOP', '0')) & 32:
    lib = capi.load()
    lib.moshii_internal_l32.restype = C.c_void_p
    lib.moshii_internal_l32.argtypes = [C.c_void_p]
    buf = (C.c_longlong * 1024)()
    lib.moshii_internal_lbs_debug_times.argtypes = [C.c_void_p, C.c_void_p]
    rc = lib.moshii_internal_lbs_debug_times(lib.moshii_internal_l32(solver.dev.handle), buf)
    tt = np.array(buf[:], dtype=np.int64).reshape(512, 2)
    t0 = tt[:, 0].min()
    st_, en_ = (tt[:, 0] - t0) * 0.01, (tt[:, 1] - t0) * 0.01    # us
    print(f'workgroups (last call): start {st_.min():.1f} .. {st_.max():.1f} us, end {en_.min():.1f} .. {en_.max():.1f} us, duration median {np.median(en_ - st_):.1f} us (min {np.min(en_ - st_):.1f}, max {np.max(en_ - st_):.1f})')
    for x in range(8):
        sel = np.arange(512) % 8 == x
        d = (en_ - st_)[sel]
        print(f'  XCD {x}: duration median {np.median(d):.1f} us, max {d.max():.1f}; first 32 slots {np.median(d[:32]):.1f}, last 32 slots {np.median(d[32:]):.1f}; latest end {en_[sel].max():.1f}')
