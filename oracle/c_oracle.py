"""TEST INFRASTRUCTURE -- ctypes loader for the scalar C oracle (dswx_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# DSWX_ORACLE_LIB selects another build of the same source (tests: the UBSan build)
LIB = os.environ.get('DSWX_ORACLE_LIB') or os.path.join(HERE, '_build', 'libdswx_oracle.so')
_lib = None


def build(force=False):
    src = os.path.join(HERE, 'dswx_oracle.c')
    if os.environ.get('DSWX_ORACLE_LIB'):
        return LIB
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(['make', '-C', HERE] + (['-B'] if force else []), check=True,
                       capture_output=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        _lib.oracle_classify.restype = ctypes.c_int
        _lib.oracle_check_quotient_predicate.restype = ctypes.c_int64
        _lib.oracle_check_quotient_predicate.argtypes = [
            ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    return _lib


def classify(params, bands, fmask, land=None, shad=None, ocean=None,
             layers=('diag', 'wtr1', 'wtr1_aerosol', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud')):
    """Same calling shape as proteus_amd._capi.Context.classify_host, on the CPU.
    `params` is a proteus_amd._capi.Params (shared struct layout, include/dswx_hip.h)."""
    from proteus_amd import _capi   # struct definitions only
    lib = load()
    bands = [np.ascontiguousarray(b, dtype=np.int16) for b in bands]
    shape = bands[0].shape
    n = bands[0].size
    keep = [bands]
    pin = _capi.PlanesIn()
    for i, b in enumerate(bands):
        pin.band[i] = b.ctypes.data
    for name, arr in (('fmask', fmask), ('land', land), ('shad', shad), ('ocean', ocean)):
        if arr is None:
            continue
        a = np.ascontiguousarray(arr, dtype=np.uint8)
        keep.append(a)
        setattr(pin, name, a.ctypes.data)
    pout = _capi.PlanesOut()
    res = {}
    for name in layers:
        dt = np.uint16 if name == 'diag' else (np.float64 if name in _capi.F64_LAYERS
                                               else np.uint8)
        res[name] = np.empty(shape, dtype=dt)
        setattr(pout, name, res[name].ctypes.data)
    cnt = np.zeros(3, dtype=np.int64)
    rc = lib.oracle_classify(ctypes.byref(params), ctypes.c_int64(n), ctypes.byref(pin),
                             ctypes.byref(pout), ctypes.c_void_p(cnt.ctypes.data))
    if rc != 0:
        raise ValueError('C oracle: unsupported mode')
    res['counters'] = cnt
    return res


def check_quotient_predicate(t, less_than, n_lo=-32768, n_hi=32768):
    lib = load()
    bn, bd = ctypes.c_int(0), ctypes.c_int(0)
    bad = lib.oracle_check_quotient_predicate(float(t), int(less_than), n_lo, n_hi,
                                              ctypes.byref(bn), ctypes.byref(bd))
    return int(bad), (bn.value, bd.value)
