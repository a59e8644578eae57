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
    assert lib.moshii_version() >= 100


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of every struct in include/moshii.h, as gcc sees them, equal the ctypes mirrors."""
    import ctypes as C
    import subprocess
    structs = {'moshii_model_desc': capi.ModelDesc, 'moshii_solve_opts': capi.SolveOpts, 'moshii_chain_desc': capi.ChainDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "moshii.h"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append('return 0;}')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
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
