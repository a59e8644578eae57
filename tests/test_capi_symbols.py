"""CPU: the C-ABI library builds/loads and exports every symbol include/moshii.h declares; without a GPU
the compute entry points fail loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from moshpp_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from moshpp_amd import build
    build.build(force=False, verbose=False)
    return capi.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, 'include', 'moshii.h')).read()
    declared = set(re.findall(r'\b(moshii_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 17
    assert declared == set(capi.EXPORTS), declared.symmetric_difference(set(capi.EXPORTS))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.moshii_version() >= 101   # 101: moshii_stagei_desc.init_sq


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of every struct in include/moshii.h, as gcc sees them, equal the ctypes mirrors."""
    import ctypes as C
    import subprocess
    structs = {'moshii_model_desc': capi.ModelDesc, 'moshii_solve_opts': capi.SolveOpts, 'moshii_chain_desc': capi.ChainDesc,
               'moshii_sequence_desc': capi.SequenceDesc, 'moshii_chunk_opts': capi.ChunkOpts,
               'moshii_chunk_report': capi.ChunkReport, 'moshii_stagei_desc': capi.StageIDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "moshii.h"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append('printf("MOSHII_NERR %d\\n", MOSHII_NERR);')
    lines.append('return 0;}')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    assert int(got['MOSHII_NERR']) == capi.NERR
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f'{cname}.{fname}']) == getattr(cls, fname).offset, (cname, fname)


def test_no_silent_cpu_fallback(lib):
    if capi.device_count() > 0:
        pytest.skip('GPU present: the fallback check is for GPU-less hosts')
    with pytest.raises(capi.MoshiiError):
        capi.require_device()
    with pytest.raises(capi.MoshiiError):
        capi.Model(np.zeros((4, 3)), np.zeros((4, 3, 1)), np.zeros((4, 3, 9)), np.ones((4, 2)) / 2, np.ones((2, 4)) / 4,
                   np.array([-1, 0]), 6)
    from moshpp_amd.chmosh import mosh_stageii
    with pytest.raises(capi.MoshiiError):
        mosh_stageii('x.npz', None, None, None, None, None)


def test_plan_chunks_is_balanced_and_covers_every_frame():
    """moshii_plan_chunks is host arithmetic (no device): chunk starts are balanced, warm-up is clipped at 0."""
    import numpy as np
    from moshpp_amd import capi
    for F, C, W in [(4000, 256, 16), (4000, 512, 16), (10, 4, 16), (3, 8, 2), (1, 1, 0), (0, 4, 16), (97, 5, 0)]:
        starts, launch = capi.plan_chunks(F, C, W)
        n = len(starts)
        assert 1 <= n <= max(1, min(C, max(F, 1)))
        assert starts[0] == 0 and launch[0] == 0
        ends = list(starts[1:]) + [F]
        lens = np.array(ends) - starts
        assert lens.sum() == F and (lens >= 0).all() and lens.max() - lens.min() <= 1
        assert (launch <= starts).all() and (starts - launch <= W).all() and (launch >= 0).all()
        assert all(launch[c] == max(0, starts[c] - W) for c in range(1, n))
    s, l = capi.plan_chunks(100, 10, 4, cap=3)
    assert len(s) == 3
