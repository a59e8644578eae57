"""TEST INFRASTRUCTURE: builds tests/emu/libmoshii_emu.so = the product's moshii_api.hip + chain_solve.hip + lbs_forward.hip + stagei.hip (UNCHANGED sources, compiled
by g++ / host clang++ against tests/emu/fakehip/hip/hip_runtime.h) + stagei.hip + the fiber scheduler.  The Stage-II chain
kernel then runs on the CPU, 256 fibers per workgroup, so its arithmetic can be held to the oracle (and run under sanitizers) without
a GPU.  Never loaded by the product; the GPU tests check the hipcc build."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'moshpp_amd', 'csrc')
OUT = os.path.join(HERE, 'libmoshii_emu.so')
# lbs_forward.hip uses clang's ext_vector_type and _Float16 arithmetic: that one unit is compiled by the ROCm clang++ as a HOST compiler
CLANGXX = '/opt/rocm/lib/llvm/bin/clang++'
UNITS = [(os.path.join(CSRC, 'moshii_api.hip'), [], 'g++'), (os.path.join(CSRC, 'chain_solve.hip'), [], 'g++'),
         (os.path.join(CSRC, 'lbs_forward.hip'), ['-DHIPEMU_NATIVE_F16'], CLANGXX),
         (os.path.join(CSRC, 'stagei.hip'), ['-DHIPEMU_UNNAMED_LDS', '-DS1_TPB=64', '-DS1_VL=8'], 'g++'), (os.path.join(HERE, 'hip_emu_runtime.cpp'), [], 'g++')]


def build(force=False, opt='-O1'):
    deps = [u for u, _, _ in UNITS] + [os.path.join(HERE, 'fakehip', 'hip', 'hip_runtime.h'), os.path.join(CSRC, 'moshii_dev.h'),
                                    os.path.join(ROOT, 'include', 'moshii.h')]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(d) for d in deps):
        return OUT
    objs, procs = [], []
    for src, extra, cxx in UNITS:
        obj = os.path.join(HERE, '_' + os.path.basename(src).replace('.', '_') + '.o')
        cmd = [cxx, '-O2' if src.endswith('stagei.hip') or src.endswith('hip_emu_runtime.cpp') else opt, '-std=c++17', '-fPIC', '-w', '-I', os.path.join(HERE, 'fakehip'), '-I', os.path.join(ROOT, 'include'), '-I', HERE,
               '-x', 'c++', '-c', src, '-o', obj] + extra
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'g++ failed on {src}')
    subprocess.check_call(['g++', '-shared', '-fPIC', '-o', OUT] + objs)
    return OUT


if __name__ == '__main__':
    print(build(force=True))
