#!/usr/bin/env python3
"""What does page-locking a caller's pageable numpy arrays in place cost (hipHostRegister / Unregister)?"""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteus_amd import _capi
ctx = _capi.Context(0)
hip = ctypes.CDLL('libamdhip64.so')
out = {}
for mb in (13, 27, 107, 281):
    a = np.zeros(mb << 20, np.uint8)
    a[::4096] = 1                      # touched
    p = ctypes.c_void_p(a.ctypes.data)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(p, ctypes.c_size_t(a.nbytes), 0)
        t1 = time.perf_counter()
        rc2 = hip.hipHostUnregister(p)
        t2 = time.perf_counter()
        ts.append((round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3), rc, rc2))
    out[f'{mb} MB'] = ts
# fresh (untouched) output array: np.empty
b = np.empty(107 << 20, np.uint8)
t0 = time.perf_counter(); rc = hip.hipHostRegister(ctypes.c_void_p(b.ctypes.data), ctypes.c_size_t(b.nbytes), 0); t1 = time.perf_counter()
hip.hipHostUnregister(ctypes.c_void_p(b.ctypes.data))
out['107 MB untouched (np.empty)'] = [round((t1 - t0) * 1e3, 3), rc]
# multi-threaded memcpy rate for comparison
import threading
src = np.ones(281 << 20, np.uint8); dst = np.empty_like(src)
def cp(i, n):
    s = len(src) // n
    dst[i * s:(i + 1) * s] = src[i * s:(i + 1) * s]
for n in (1, 4, 8, 16):
    for rep in range(2):
        th = [threading.Thread(target=cp, args=(i, n)) for i in range(n)]
        t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t1 = time.perf_counter()
    out[f'memcpy 281 MB, {n} threads (numpy slices release the GIL), 2nd pass'] = round((t1 - t0) * 1e3, 2)
print(json.dumps(out, indent=1))
