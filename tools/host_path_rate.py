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
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
