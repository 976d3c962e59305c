#!/usr/bin/env python3
"""Interleaved A/B of kernel variants in ONE process on ONE device (one context per variant, each
configured through libdswx_lab.so's switches), as cdna_hip_programming.md rule 24 asks.

    python tools/ab_variants.py --tiles 64 --rounds 7 fused_variant=3 fused_variant=0 fused_variant=2 ...
    python tools/ab_variants.py tune_lut_wps=4 tune_lut_wps=5,fused_variant=3
    python tools/ab_variants.py auto LIB=proteus_amd/_lib/ab/libdswx_prev.so   # two builds of the product library
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
    ap.add_argument('variants', nargs='+', help='key=value[,key=value] per variant (lab switches), or auto')
    ap.add_argument('--tiles', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--masks', action='store_true')
    ap.add_argument('--tile-align', type=int, default=256)
    ap.add_argument('--mode', default='mask', choices=['mask', 'ignore', 'cover'], help="mask_adjacent_to_cloud_mode ('cover': the three-kernel path)")
    a = ap.parse_args()
    ctxs = []
    for v in a.variants:
        lib_path, settings = None, {}
        for kv in v.split(','):
            if kv == 'auto':
                continue
            k, val = kv.split('=')
            if k == 'LIB':              # another build of the product library (same ABI), e.g. the previous commit
                lib_path = os.path.abspath(val)
            else:
                settings[k] = int(val)
        ctx = _capi.Context(0, lib_path=lib_path)
        if settings:
            ctx.lab_configure(**settings)
        ctxs.append(ctx)
    base = ctxs[0]
    batch = _capi.DeviceBatch(base, a.tiles, 3660, 3660, masks=a.masks, tile_align=a.tile_align)
    batch.synth(SEED)
    base.synchronize()
    p = _capi.make_params(mask_adjacent_to_cloud_mode=a.mode)
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
