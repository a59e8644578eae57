"""How many joints does a run of consecutive vertices depend on?  (the lane = vertex LBS design of DESIGN.md's next steps needs
few: the joints of a 64-vertex group would be scalar operands.)  Synthetic SMPL-H / SMPL-X bodies in both vertex orders.  CPU only."""
import sys
sys.path.insert(0, '.')
import numpy as np
from moshpp_amd import synth

for mt in ('smplh', 'smplx'):
    for order in ('shuffled', 'mesh'):
        dd = synth.synth_model(mt, seed=1000, vertex_order=order)
        W = dd['weights'] != 0
        V = W.shape[0]
        line = f'{mt} {order:>8}: influences per vertex mean {W.sum(1).mean():.2f} |'
        for g in (16, 64, 128):
            n = np.array([W[s:s + g].any(0).sum() for s in range(0, V, g)])
            line += f' joints per {g:3d} consecutive vertices: mean {n.mean():5.1f} median {int(np.median(n)):2d} p90 {int(np.percentile(n, 90)):2d} max {n.max():2d} |'
        print(line)
