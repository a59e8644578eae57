"""Writes the C3D on-disk variants our reader must ingest -- MIPS (big-endian) float, DEC float, Intel scaled-int16, MIPS
scaled-int16 -- with OUR writer, and parses each with the REFERENCE's vendored reader (/root/reference/src/moshpp/tools/
c3d.py `Reader`, :35-60 processor dtypes, :1293-1385 frame decoding) to record what an independent implementation
makes of them.  The test then requires our reader to agree with that record.

The reference reader trips over NumPy 2 in its float branch (`last_word & 0x80008000` on an int32 array, c3d.py:1333);
the module source is patched IN MEMORY for that one expression (the file on disk is untouched, nothing is copied).
Run in the build container only; the .c3d files and the npz are committed."""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from moshpp_amd import c3d_io  # noqa: E402

src = open('/root/reference/src/moshpp/tools/c3d.py').read()
needle = '(last_word & 0x80008000) == 0'
assert src.count(needle) == 1
src = src.replace(needle, '(last_word.astype(np.int64) & 0x80008000) == 0')
ref = types.ModuleType('ref_c3d')
exec(compile(src, 'ref_c3d.py', 'exec'), ref.__dict__)

rng = np.random.default_rng(20260927)
F, N = 6, 41
pts = rng.normal(0, 400, (F, N, 3))
pts[rng.random((F, N)) < 0.12] = np.nan
labels = [f'MK{i:02d}' for i in range(N)]
VARIANTS = {'mips_float': (c3d_io.PROC_MIPS, None), 'dec_float': (c3d_io.PROC_DEC, None),
            'intel_int': (c3d_io.PROC_INTEL, 0.1), 'mips_int': (c3d_io.PROC_MIPS, 0.1)}
out = {'points_written': pts, 'labels': np.array(labels)}
for name, (proc, isc) in VARIANTS.items():
    fn = os.path.join(HERE, f'variant_{name}.c3d')
    c3d_io.write_c3d(fn, pts, labels, frame_rate=100.0, processor=proc, int_scale=isc)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with open(fn, 'rb') as h:
            r = ref.Reader(h)
            frames = [p.copy() for _, p, _ in r.read_frames()]
            lab = [l.strip() for l in r.point_labels]
            rate = float(r.point_rate)
    arr = np.array(frames)                       # [F, N, 5]: x, y, z, residual, cameras (-1 / -1 = invalid)
    out[f'{name}_xyz'] = arr[:, :, :3]
    out[f'{name}_invalid'] = arr[:, :, 3] < 0
    out[f'{name}_rate'] = rate
    assert lab == labels, (name, lab[:3])
    print(name, os.path.getsize(fn), 'bytes; reference reader: max |xyz - written| =',
          float(np.nanmax(np.abs(np.where(out[f'{name}_invalid'][..., None], np.nan, arr[:, :, :3]) - pts))))
np.savez_compressed(os.path.join(HERE, 'c3d_variants_expected.npz'), **out)
