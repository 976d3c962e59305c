#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry (dswx_classify_host) and the
device-resident single-tile rate, for DESIGN.md §6 (never the bench `value`)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteus_amd import _capi                  # noqa: E402
from proteus_amd.synth import SEED             # noqa: E402


def main():
    ctx = _capi.Context(0)
    T = 3660
    out = {}
    # device-resident single tile (BASELINE configs[1]); fits the 256 MiB Infinity Cache
    b1 = _capi.DeviceBatch(ctx, 1, T, T)
    b1.synth(SEED)
    p = _capi.default_params()
    for _ in range(3):
        b1.classify(p)
    ctx.synchronize()
    ms = []
    for _ in range(20):
        a, b = ctx.event(), ctx.event()
        ctx.record(a); b1.classify(p); ctx.record(b)
        ms.append(ctx.elapsed_ms(a, b))
    out['single_tile_device_resident'] = {
        'ms_avg': sum(ms) / len(ms), 'ms_min': min(ms),
        'Mpix_s': T * T / (sum(ms) / len(ms)) / 1e3,
        'note': 'one 3660^2 tile repeatedly: 281 MB working set, partly served by the 256 MiB '
                'Infinity Cache; includes the counters memset + finishing kernel'}
    # host-pointer path, pageable numpy arrays, one tile
    bands = [b1.read_tile(n, 0) for n in _capi.BAND_NAMES]
    fmask = b1.read_tile('fmask', 0)
    ctx.classify_host(bands, fmask, p)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.classify_host(bands, fmask, p, layers=('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'))
        ts.append(time.perf_counter() - t0)
    out['host_pointer_path'] = {
        's_avg': sum(ts) / len(ts), 's_min': min(ts), 'Mpix_s': T * T / (sum(ts) / len(ts)) / 1e6,
        'GBps_moved': T * T * 21 / (sum(ts) / len(ts)) / 1e9,
        'note': 'dswx_classify_host from pageable host memory: H2D of 174 MB + kernel + D2H of '
                '107 MB, synchronous, includes numpy output allocation'}
    # the same from page-locked arrays: zero copy (the kernels work on the host planes across PCIe)
    pb = []
    for a in bands + [fmask]:
        q = ctx.pinned_empty(a.shape, a.dtype)
        q[...] = a
        pb.append(q)
    ctx.classify_host(pb[:6], pb[6], p)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = ctx.classify_host(pb[:6], pb[6], p, layers=('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'))
        ts.append(time.perf_counter() - t0)
        del res
    out['host_pointer_path_pinned'] = {
        's_avg': sum(ts) / len(ts), 's_min': min(ts), 'Mpix_s': T * T / (sum(ts) / len(ts)) / 1e6,
        'GBps_moved': T * T * 21 / (sum(ts) / len(ts)) / 1e9, 'kernel': ctx.last_kernel_info(),
        'note': 'dswx_classify_host from page-locked host memory (dswx_host_alloc): zero copy; includes allocating '
                'page-locked outputs'}
    # and with the page-locked output allocation taken out (outputs reused): raw library call
    import ctypes
    pin, pout = _capi.PlanesIn(), _capi.PlanesOut()
    for i in range(6):
        pin.band[i] = pb[i].ctypes.data
    pin.fmask = pb[6].ctypes.data
    outs = {'diag': ctx.pinned_empty((T, T), np.uint16)}
    for n in ('wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
        outs[n] = ctx.pinned_empty((T, T), np.uint8)
    for n, a in outs.items():
        setattr(pout, n, a.ctypes.data)
    cnt = np.zeros((1, 3), np.int64)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        rc = ctx.lib.dswx_classify_host(ctx.handle, ctypes.byref(p), 1, T, T, ctypes.byref(pin), ctypes.byref(pout),
                                        ctypes.c_void_p(cnt.ctypes.data))
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    ts = ts[1:]
    out['host_pointer_path_pinned_reused_outputs'] = {
        's_avg': sum(ts) / len(ts), 's_min': min(ts), 'Mpix_s': T * T / (sum(ts) / len(ts)) / 1e6,
        'GBps_moved': T * T * 21 / (sum(ts) / len(ts)) / 1e9}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
