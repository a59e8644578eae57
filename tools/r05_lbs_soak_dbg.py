"""Round-5 debugging aid: are the wrong floats of the export NEVER WRITTEN (sentinel survives) or written wrong?"""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from collections import Counter
from tests.helpers import oracle_case, device_case
from moshpp_amd import synth
case = oracle_case('smplh', F=4, M=53, seed=77, dd=synth.synth_model('smplh', seed=77, vertex_order=os.environ.get('LBS_BODY', 'shuffled')))
dev = device_case(case)
m = case['m']
rng = np.random.default_rng(2026)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 701
pose = rng.normal(0, 0.4, (F, m['NP']))
trans = rng.normal(0, 1.5, (F, 3))
ref = dev['model'].lbs_forward(pose, trans)
d = torch.device('cuda', 0)
p32 = torch.from_numpy(pose.astype(np.float32)).to(d); t32 = torch.from_numpy(trans.astype(np.float32)).to(d)
V = ref.shape[1]
for rep in range(3):
    out = torch.full((F, V, 3), 7.0e30, dtype=torch.float32, device=d)
    torch.cuda.synchronize()
    dev['model'].lbs_forward_device(F, p32.data_ptr(), t32.data_ptr(), out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    g = out.cpu().numpy()
    sent = (g == np.float32(7.0e30))
    bad = (np.abs(g - ref) > 2e-5) | np.isnan(g)
    print(f'rep {rep}: wrong floats {bad.sum()}, of which never written {int((bad & sent).sum())}; sentinel floats {sent.sum()}')
    if bad.any():
        fr, vv, cc = np.nonzero(bad)
        fl_idx = (vv * 3 + cc) % 192        # float index inside a 64-vertex tile row
        print('   float-in-row piece columns (piece = 4 floats):', sorted(Counter(int(x) // 4 for x in fl_idx).items())[:30])
        print('   rows-in-block:', sorted(Counter(int(x) % 16 for x in fr).items()), ' blocks:', sorted(Counter((int(x) % 128) // 16 for x in fr).items()))
        k = (fr % 16) * 48 + fl_idx // 4
        print('   piece numbers k = 48 row + piece -> thread (k % 256), s (k // 256):', sorted(Counter((int(x) % 256, int(x) // 256) for x in k).items())[:24])
