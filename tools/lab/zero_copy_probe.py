#!/usr/bin/env python3
"""Host-pointer entry from page-locked planes: zero copy (product) vs the staged three-stream pipeline (lab
switch host_pipeline=1), the C entry timed with its outputs allocated beforehand.

    python tools/lab/zero_copy_probe.py [tiles]
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from proteus_amd import _capi            # noqa: E402
from proteus_amd.synth import synth_tile  # noqa: E402

T = 3660
LAYERS = ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud')


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    s = synth_tile(3, T, T)
    out = {'tiles': n}
    results = {}
    for mode, switch in (('zero_copy', None), ('staged_pipeline_8_chunks', {'host_pipeline': 1})):
        ctx = _capi.Context(0)
        if switch:
            ctx.lab_configure(**switch)
        p = _capi.default_params()
        bands = [ctx.pinned_empty((n, T, T), np.int16) for _ in range(6)]
        fm = ctx.pinned_empty((n, T, T), np.uint8)
        for t in range(n):
            for i in range(6):
                bands[i][t] = s['bands'][i]
            fm[t] = s['fmask']
        outs = {k: ctx.pinned_empty((n, T, T), np.uint16 if k == 'diag' else np.uint8) for k in LAYERS}
        pin, pout = _capi.PlanesIn(), _capi.PlanesOut()
        for i in range(6):
            pin.band[i] = bands[i].ctypes.data
        pin.fmask = fm.ctypes.data
        for k in LAYERS:
            setattr(pout, k, outs[k].ctypes.data)
        cnt = np.zeros((n, 3), np.int64)
        times = []
        for rep in range(5):
            t0 = time.perf_counter()
            _capi._check(ctx.lib.dswx_classify_host(ctx.handle, ctypes.byref(p), n, T, T, ctypes.byref(pin),
                                                    ctypes.byref(pout), _capi._host_ptr(cnt)))
            times.append(time.perf_counter() - t0)
        best = min(times[1:])
        out[mode] = {'ms': round(best * 1e3, 2), 'Gpx_s': round(n * T * T / best / 1e9, 3),
                     'GBps_in_plus_out': round(n * T * T * 21 / best / 1e9, 1), 'kernel': ctx.last_kernel_info()}
        results[mode] = {k: np.array(v) for k, v in outs.items()}
        results[mode]['counters'] = cnt.copy()
        del bands, fm, outs
        ctx.close()
    # pageable planes (plain numpy arrays, as a caller of the reference's seam holds them): synchronous staged copies
    ctx = _capi.Context(0)
    p = _capi.default_params()
    bands = [np.ascontiguousarray(np.broadcast_to(s['bands'][i], (n, T, T))) for i in range(6)]
    fm = np.ascontiguousarray(np.broadcast_to(s['fmask'], (n, T, T)))
    times = []
    for rep in range(4):
        t0 = time.perf_counter()
        res = ctx.classify_host(bands, fm, p, layers=LAYERS)
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    out['pageable_synchronous_copies'] = {'ms': round(best * 1e3, 2), 'Gpx_s': round(n * T * T / best / 1e9, 3), 'kernel': ctx.last_kernel_info(),
                                          'note': 'through the Python wrapper: includes allocating the numpy outputs'}
    out['pageable_identical'] = all(np.array_equal(res[k], results['zero_copy'][k]) for k in LAYERS)
    ctx.close()
    a, b = list(results.values())[:2]
    out['identical'] = all(np.array_equal(a[k], b[k]) for k in a)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
