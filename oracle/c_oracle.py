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


def _digest():
    import hashlib
    h = hashlib.sha256()
    for path in (os.path.join(HERE, 'dswx_oracle.c'), os.path.join(HERE, 'Makefile'),
                 os.path.join(os.path.dirname(HERE), 'include', 'dswx_hip.h')):
        with open(path, 'rb') as f:
            h.update(f.read())
        h.update(b'\0')
    return h.hexdigest()


def _fresh():
    try:
        with open(LIB + '.srchash') as f:
            return os.path.exists(LIB) and f.read().strip() == _digest()
    except OSError:
        return False


def build(force=False):
    """Compile the C oracle when the stamp beside it does not hold the digest of its sources (content, not mtime: a
    copied tree reorders mtimes).  Safe when many processes call it at once -- every rank of `bench.py --gpus 8` checks
    its tiles with it: one builder under a file lock, the library appears by an atomic rename."""
    if os.environ.get('DSWX_ORACLE_LIB'):
        return LIB
    if not force and _fresh():
        return LIB
    import fcntl
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(os.path.join(os.path.dirname(LIB), '.build.lock'), 'a') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or not _fresh():
                digest = _digest()
                tmp = f'_build/libdswx_oracle.so.{os.getpid()}.tmp'
                subprocess.run(['make', '-C', HERE, '-B', f'OUT={tmp}'], check=True, capture_output=True)
                if os.path.exists(LIB + '.srchash'):
                    os.remove(LIB + '.srchash')
                os.replace(os.path.join(HERE, tmp), LIB)
                with open(LIB + '.srchash', 'w') as f:
                    f.write(digest + '\n')
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
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
