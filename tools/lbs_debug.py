"""Where does the f32 export differ from the f64 kernel?  python tools/lbs_debug.py smplh:300 smplx:150"""
import gc
import sys
import numpy as np
sys.path.insert(0, '.')
from tests.helpers import oracle_case, device_case
for spec in sys.argv[1:] or ['smplx:150']:
    mt, F = spec.split(':'); F = int(F)
    M = {'smplh': 53, 'smplx': 60, 'mano': 24, 'smpl': 41}[mt]
    case = oracle_case(mt, F=4, M=M, seed=61)
    dev = device_case(case)
    m = case['m']
    rng = np.random.default_rng(5)
    pose = rng.normal(0, 0.35, (F, m['NP']))
    trans = rng.normal(0, 1, (F, 3))
    ref = dev['model'].lbs_forward(pose, trans)
    got = dev['model'].lbs_forward(pose, trans, dtype=np.float32)
    err = np.abs(got - ref).max(-1)          # [F, V]
    print(spec, 'max', err.max(), 'rms', np.sqrt((err ** 2).mean()))
    bad = np.argwhere(err > 2e-5)
    print('bad (frame, vertex) pairs:', len(bad), 'of', err.size)
    if len(bad):
        fs, vs = np.unique(bad[:, 0]), np.unique(bad[:, 1])
        print('frames', fs[:40], '... n', len(fs))
        print('verts', vs[:40], '... n', len(vs), 'min', vs.min(), 'max', vs.max())
        w = case['model']['weights']
        print('influences of first bad verts:', [np.flatnonzero(w[v]).tolist() for v in vs[:6]])
        print('err by frame%8:', [float(err[f::8][:, vs].max()) for f in range(8)])
        f0b, v0b = bad[0]
        d = np.abs(ref[:, v0b] - got[f0b, v0b]).max(-1)
        print('bad sample', f0b, v0b, 'got', got[f0b, v0b], 'ref', ref[f0b, v0b], 'closest ref frame for this vertex:', int(d.argmin()), float(d.min()))
        d2 = np.abs(ref[f0b] - got[f0b, v0b]).max(-1)
        print('closest ref vertex in the same frame:', int(d2.argmin()), float(d2.min()))
        print('bad v%32 histogram:', np.bincount(vs % 32, minlength=32).tolist())
    del dev, case
    gc.collect()
