#!/usr/bin/env python3
"""Interleaved A/B of kernel variants in ONE process on ONE device (contexts created with
different DSWX_* environment knobs), as cdna_hip_programming.md rule 24 asks.

    python tools/ab_variants.py --tiles 64 --rounds 7 DSWX_TUNE_WPS=4 DSWX_TUNE_WPS=6 ...
    python tools/ab_variants.py DSWX_TUNE_WPS=4 LIB=proteus_amd/_lib/ab/libdswx_prev.so   # two builds
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteus_amd import _capi            # noqa: E402
from proteus_amd.synth import SEED       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('variants', nargs='+', help='ENV=VALUE[,ENV=VALUE] per variant')
    ap.add_argument('--tiles', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--masks', action='store_true')
    ap.add_argument('--tile-align', type=int, default=256)
    a = ap.parse_args()
    ctxs = []
    for v in a.variants:
        saved = {}
        lib_path = None
        for kv in v.split(','):
            k, val = kv.split('=')
            if k == 'LIB':              # another build of the library (same ABI), e.g. the previous commit
                lib_path = os.path.abspath(val)
                continue
            saved[k] = os.environ.get(k)
            os.environ[k] = val
        ctxs.append(_capi.Context(0, lib_path=lib_path))
        for k, old in saved.items():
            if old is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = old
    base = ctxs[0]
    batch = _capi.DeviceBatch(base, a.tiles, 3660, 3660, masks=a.masks, tile_align=a.tile_align)
    batch.synth(SEED)
    base.synchronize()
    p = _capi.default_params()
    px = a.tiles * 3660 * 3660
    bpp = 24 if a.masks else 21
    res = {v: [] for v in a.variants}
    info = {}
    for r in range(a.rounds):
        for v, ctx in zip(a.variants, ctxs):
            ctx.classify_batch(p, batch.geom, batch.pin, batch.pout, batch.counters_ptr)
            ctx.synchronize()
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0)
            for _ in range(a.reps):
                ctx.classify_batch(p, batch.geom, batch.pin, batch.pout, batch.counters_ptr)
            ctx.record(e1)
            ms = ctx.elapsed_ms(e0, e1) / a.reps
            res[v].append(px * bpp / ms / 1e6)
            info[v] = ctx.last_kernel_info()
    out = {v: {'GBps_median': round(statistics.median(x), 1), 'GBps_min': round(min(x), 1),
               'GBps_max': round(max(x), 1), 'kernel': info[v]} for v, x in res.items()}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
