"""TEST INFRASTRUCTURE: drive the Stage-I solver of the CPU emulation build (tests/emu/build_chain_emu.py: stagei.hip compiled unchanged
against the stand-in HIP runtime, kernels on fibers) from NumPy arrays."""
import ctypes as C

import numpy as np

from moshpp_amd import capi
from . import build_emu

CORE_SYMBOL = '_Z18moshii_stagei_corePK11S1ModelViewPK11S1PriorViewPK18moshii_stagei_descPvPci'


class ModelView(C.Structure):
    _fields_ = [('V', C.c_int), ('K', C.c_int), ('NB', C.c_int), ('NP', C.c_int), ('body_dof', C.c_int), ('hand_dof', C.c_int),
                ('parents', C.c_void_p), ('anc', C.c_void_p), ('vt', C.c_void_p), ('shapedirs', C.c_void_p), ('posedirs', C.c_void_p),
                ('weights', C.c_void_p), ('Jreg', C.c_void_p), ('hands_mean', C.c_void_p), ('comps', C.c_void_p)]


class PriorView(C.Structure):
    _fields_ = [('G', C.c_int), ('npose', C.c_int), ('means', C.c_void_p), ('chols', C.c_void_p), ('neglogw', C.c_void_p)]


def solve(m, prior, **kw):
    """m: oracle prepare_model() dict; prior: oracle prepare_gmm_prior() dict or None; kw: capi.stagei_desc arguments."""
    lib = C.CDLL(build_emu.build())
    keep = []

    def ptr(a, dt):
        a = np.ascontiguousarray(a, dtype=dt); keep.append(a)
        return a.ctypes.data
    K = m['K']
    anc = np.zeros(K, dtype=np.uint64)
    for k in range(K):
        for j in range(K):
            if m['anc'][k, j]:
                anc[k] |= np.uint64(1) << np.uint64(j)
    mv = ModelView(m['v_template'].shape[0], K, m['shapedirs'].shape[2], m['NP'], m['body_dof'], m['hand_dof'],
                   ptr(m['parents'], np.int32), ptr(anc, np.uint64), ptr(m['v_template'], np.float64), ptr(m['shapedirs'], np.float64),
                   ptr(m['posedirs'], np.float64), ptr(m['weights'], np.float64), ptr(m['J_regressor'], np.float64),
                   ptr(m['hands_mean'], np.float64) if m['hand_dof'] else None,
                   ptr(m['selected_components'], np.float64) if m['hand_dof'] else None)
    pv = None
    if prior is not None:
        pv = PriorView(len(prior['weights']), prior['npose'], ptr(prior['means'], np.float64), ptr(prior['chols'], np.float64),
                       ptr(-np.log(prior['weights']), np.float64))
    desc, out, keep2 = capi.stagei_desc(NP=m['NP'], **kw)
    err = C.create_string_buffer(256)
    core = getattr(lib, CORE_SYMBOL)     # moshii_stagei_core(mv, pv, desc, stream, err, errlen): C++ linkage in stagei_views.h
    core.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    core.restype = C.c_int
    rc = core(C.byref(mv), C.byref(pv) if pv is not None else None, C.byref(desc), None, err, 256)
    if rc != 0:
        raise RuntimeError(f'stagei emu failed ({rc}): {err.value.decode()}')
    return out
