"""In-kernel phase timing of the extended chain kernel on the config-3 shape (SMPL-X, 89 markers, fingers + jaw + E
expression coefficients).  Needs the -DMOSHII_PROFILE build: MOSHII_LIB=moshpp_amd/libmoshii_prof.so
python tools/prof_config3.py [F=40] [E=80]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from moshpp_amd import capi                                    # noqa: E402
from tests.helpers import shape_case, device_case              # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 40
E = int(sys.argv[2]) if len(sys.argv) > 2 else 80
case = shape_case('smplx', F=F, M=89, E=E, seed=21, kind='expr')
dev = device_case(case, optimize_fingers=True, optimize_face=True, shape_kind='expr')
lib = capi.load()
buf = (C.c_longlong * 64)()
lib.moshii_prof_read.argtypes = [C.POINTER(C.c_longlong), C.c_int]
ch = [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)]
capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], [dict(ch[0], obs=case['obs'][:3], vis=case['vis'][:3])])
lib.moshii_prof_read(buf, 1)
t = time.perf_counter(); out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], ch)[0]; dt = time.perf_counter() - t
lib.moshii_prof_read(buf, 1)
p = np.array(list(buf), dtype=np.float64)
names = {0: 'eval: shape+fullpose/rodrigues/chain', 1: 'eval: posedirs', 2: 'eval: skin+markers', 3: 'eval: prior+reduce',
         4: 'asm: pre-phase (+ shape chain) + T0', 5: 'asm: T1 vertex jac + T1s shape', 6: 'asm: T2 marker rows',
         7: 'asm: T3 JtJ', 8: 'asm: structured', 9: 'ldl: setup/tail', 13: 'ldl: U2 (last panel, scalar) + publish', 14: 'ldl: 16-column elimination (wave 0) || U1 (MFMA, waves 1-3)',
         10: 'back-subst', 31: 'coop: exchange after an evaluation (incl. waiting)', 32: 'coop: exchange of the normal equations (incl. waiting)',
         12: 'kernel total'}
tot = p[12]
print(f'smplx E={E} F={F} wall {dt*1e3:.1f} ms ({dt/F*1e6:.1f} us/frame) launch {capi.last_launch_info()}')
print(f'evals {p[20]/F:.2f}/frame assembles {p[21]/F:.2f}/frame ldl {p[22]/F:.2f}/frame')
us_per_tick = (p[30] / 100e6) * 1e6 / tot
acc = 0
for k, nm in names.items():
    if k == 12:
        continue
    acc += p[k]
    print(f'  {nm:40s} {p[k]/tot*100:6.2f}%  {p[k]*us_per_tick/F:9.1f} us/frame')
print(f'  {"other (dogleg control, copies)":40s} {(tot-acc)/tot*100:6.2f}%  {(tot-acc)*us_per_tick/F:9.1f} us/frame')
