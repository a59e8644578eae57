import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gpu_lib():
    from moshpp_amd import capi
    if os.environ.get('MOSHII_EMULATE') == '1':
        # development aid: run the (small) GPU parity tests against the CPU emulation build of the same sources, e.g.
        #   MOSHII_EMULATE=1 python -m pytest tests/test_gpu_parity.py -m gpu -k 'not lbs'      (slow there)
        from tests.emu import build_chain_emu
        capi.LIB_PATH, capi._lib = build_chain_emu.build(), None
    lib = capi.load()
    if capi.device_count() < 1:
        pytest.fail('GPU test selected but no HIP device is visible (no CPU fallback exists)')
    return lib
