"""TEST INFRASTRUCTURE: point moshpp_amd.capi at the CPU emulation build of libmoshii (tests/emu/build_chain_emu.py) for the duration of
a test.  The Stage-II kernels then run as fibers on the CPU (unchanged product sources)."""
import contextlib

from moshpp_amd import capi
from . import build_chain_emu


@contextlib.contextmanager
def emulated_libmoshii():
    path = build_chain_emu.build()
    saved = (capi.LIB_PATH, capi._lib)
    capi.LIB_PATH, capi._lib = path, None
    try:
        yield capi
    finally:
        capi.LIB_PATH, capi._lib = saved
