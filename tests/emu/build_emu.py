"""TEST INFRASTRUCTURE: the Stage-I kernels run in the same CPU emulation build as the Stage-II ones (build_chain_emu.py: the .hip sources
unchanged against tests/emu/fakehip, every workgroup as fibers); this module keeps the name the tests build through."""
from .build_chain_emu import build  # noqa: F401

if __name__ == '__main__':
    print(build(force=True))
