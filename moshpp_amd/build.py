"""Build libmoshii.so in-tree for gfx950:  python -m moshpp_amd.build [--force]"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['moshii_api.hip', 'chain_solve.hip', 'lbs_forward.hip', 'stagei.hip']
# per-file extras.  chain_solve: the iterative-ILP machine scheduler orders the long dependent f64 chains of the solver better than the
# default (measured 393 vs 414 us/frame on the bench sequence, same results); the LBS kernel pins its own order with sched_barriers
# and keeps the default scheduler.
EXTRA_FLAGS = {'chain_solve.hip': ['-mllvm', '-amdgpu-sched-strategy=iterative-ilp'],
               # lbs_forward: the SLP vectoriser packs the blend's scalar FMAs into v_pk_fma_f32 and pays for it with a v_mov per operand
               # pair (a third of the epilogue's VALU instructions were moves)
               'lbs_forward.hip': ['-fno-slp-vectorize']}
if os.environ.get('MOSHII_NO_ILP'):
    EXTRA_FLAGS = {}
HEADERS = ['moshii_dev.h', 'stagei_views.h', os.path.join('..', '..', 'include', 'moshii.h')]
OUT = os.path.join(HERE, 'libmoshii.so')
LAST_BUILD_RAN_HIPCC = False   # set by build(): did this call compile anything?


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return 'hipcc'


def source_hash():
    """First 16 hex digits of the SHA-256 over the native sources and headers (names and bytes).  Compiled into the library
    (moshii_source_hash(), -DMOSHII_SRC_HASH) and written beside it, so that a stale binary -- one that was not built from this
    tree, whatever its time stamp says -- is noticed (needs_build below, tests/test_gpu_parity.py::test_loaded_library_was_built_from_this_tree)."""
    import hashlib
    h = hashlib.sha256()
    for rel in sorted(SOURCES) + sorted(HEADERS):
        fn = os.path.normpath(os.path.join(CSRC, rel))
        h.update(os.path.basename(fn).encode())
        with open(fn, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(OUT):
        return True
    try:
        with open(OUT + '.srchash') as fh:
            if fh.read().strip() != source_hash():
                return True
    except OSError:
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True, profile=False, variant=None, defines=()):
    """profile=True builds libmoshii_prof.so with in-kernel clock64() phase laps (tools/prof_chain.py).
    variant / defines: development builds libmoshii_<variant>[_prof].so with extra -D flags (kernel experiments; selected at run
    time with MOSHII_LIB)."""
    tag = ('_' + variant if variant else '') + ('_prof' if profile else '')
    out = OUT.replace('libmoshii.so', f'libmoshii{tag}.so')
    global LAST_BUILD_RAN_HIPCC
    LAST_BUILD_RAN_HIPCC = False
    if not force and not tag and not needs_build():
        return OUT
    LAST_BUILD_RAN_HIPCC = True
    shash = source_hash()
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, s.replace('.hip', f'{tag}.o'))
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-Wno-unused-value',
               '-c', os.path.join(CSRC, s), '-o', obj] + (['-DMOSHII_PROFILE'] if profile else []) + list(defines) + EXTRA_FLAGS.get(s, []) + [f'-DMOSHII_SRC_HASH="{shash}"']
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(obj)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {s}')
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(out + '.srchash', 'w') as fh:
        fh.write(shash + '\n')
    return out


if __name__ == '__main__':
    _var = [a.split('=', 1)[1] for a in sys.argv if a.startswith('--variant=')]
    print(build(force='--force' in sys.argv, profile='--profile' in sys.argv, variant=_var[0] if _var else None,
                defines=[a for a in sys.argv[1:] if a.startswith('-D')]))
