"""TEST INFRASTRUCTURE: compiles moshpp_amd/csrc/stagei.hip with g++ (-DS1_EMU: every kernel body runs as one sequential
"thread" per block, HIP calls become malloc/memcpy) so that the arithmetic of the Stage-I kernels and of the host-side dogleg
can be checked against the oracle on a CPU-only machine.  Never imported by the product; the GPU tests check the real build."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, 'moshpp_amd', 'csrc', 'stagei.hip')
OUT = os.path.join(HERE, 'libstagei_emu.so')


def build(force=False):
    deps = [SRC, os.path.join(ROOT, 'moshpp_amd', 'csrc', 'stagei_views.h'), os.path.join(ROOT, 'include', 'moshii.h'),
            os.path.join(HERE, 'emu_entry.cpp'), os.path.join(HERE, 'stagei_emu_twins.h')]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(d) for d in deps):
        return OUT
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-DS1_EMU', '-x', 'c++', SRC,
                           os.path.join(HERE, 'emu_entry.cpp'), '-o', OUT, '-I', os.path.join(ROOT, 'include'), '-I', HERE])
    return OUT


if __name__ == '__main__':
    print(build(force=True))
