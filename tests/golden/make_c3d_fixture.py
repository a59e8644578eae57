"""Generates tests/golden/refwriter_6x41.c3d with the reference's OWN vendored C3D writer
(/root/reference/src/moshpp/tools/c3d.py:1396-1608, py-c3d) plus the arrays it was fed, so that
moshpp_amd.c3d_io.read_c3d is checked against an independent implementation of the format.
Run in the build container only (needs /root/reference); the outputs are committed."""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('ref_c3d', '/root/reference/src/moshpp/tools/c3d.py')
ref_c3d = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_c3d)

rng = np.random.default_rng(20260925)
F, N = 6, 41
pts = rng.normal(0, 500, (F, N, 3)).astype(np.float32)        # millimetres
invalid = rng.random((F, N)) < 0.1
labels = [f'M{i:02d}' for i in range(N)]
labels[3] = 'subj:LFHD'
labels[7] = 'R SHO'
frames = []
for f in range(F):
    p = np.zeros((N, 5), dtype=np.float32)
    p[:, :3] = pts[f]
    p[invalid[f], :3] = 0.0
    p[:, 3] = np.where(invalid[f], -1.0, 0.0)   # residual: -1 marks an invalid sample
    p[:, 4] = np.where(invalid[f], -1.0, 0.0)   # camera mask column; py-c3d writes a negative 4th word when either is < 0
    frames.append((p, np.zeros((0, 0), dtype=np.float32)))
w = ref_c3d.Writer(point_rate=100.0)
w.add_frames(frames)
out = os.path.join(HERE, 'refwriter_6x41.c3d')
with open(out, 'wb') as h:
    w.write(h, labels)
np.savez(os.path.join(HERE, 'refwriter_6x41_expected.npz'), points=pts, invalid=invalid, labels=np.array(labels),
         rate=100.0)
print(out, os.path.getsize(out))
