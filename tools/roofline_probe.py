#!/usr/bin/env python3
"""Times the stream probe (same bytes as the fused kernel, trivial math) next to the
fused classify kernel on one device-resident batch; prints GB/s for both.

    python tools/roofline_probe.py [--tiles 64] [--reps 10] [--masks]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteus_amd import _capi            # noqa: E402
from proteus_amd.synth import SEED       # noqa: E402


def timed(ctx, fn, reps):
    fn()
    ctx.synchronize()
    ms = []
    for _ in range(reps):
        a, b = ctx.event(), ctx.event()
        ctx.record(a)
        fn()
        ctx.record(b)
        ms.append(ctx.elapsed_ms(a, b))
        ctx.destroy_event(a)
        ctx.destroy_event(b)
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tiles', type=int, default=64)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--masks', action='store_true')
    ap.add_argument('--size', type=int, default=3660)
    a = ap.parse_args()
    ctx = _capi.Context(0)
    batch = _capi.DeviceBatch(ctx, a.tiles, a.size, a.size, masks=a.masks)
    batch.synth(SEED)
    ctx.synchronize()
    p = _capi.default_params()
    px = a.tiles * a.size * a.size
    out = {'tiles': a.tiles, 'pixels': px}
    ms = timed(ctx, lambda: batch.classify(p), a.reps)
    bpp = 24 if a.masks else 21
    out['classify'] = {'kernel': ctx.last_kernel_info(), 'ms_avg': sum(ms) / len(ms), 'ms_min': min(ms),
                       'GBps_avg': px * bpp / (sum(ms) / len(ms)) / 1e6,
                       'Gpix_s': px / (sum(ms) / len(ms)) / 1e6}
    ms = timed(ctx, lambda: batch.classify(p, counters=False), a.reps)
    out['classify_nocounters'] = {'ms_avg': sum(ms) / len(ms), 'GBps_avg': px * bpp / (sum(ms) / len(ms)) / 1e6}
    if not a.masks:
        for variant in (256, 258):
            ms = timed(ctx, lambda: ctx.stream_probe(a.tiles, batch.n_pixels, batch.pin, batch.pout,
                                                     variant), a.reps)
            out[f'flat 2-stream copy nt={int(bool(variant & 2))}'] = round(px * 21 / (sum(ms) / len(ms)) / 1e6, 1)
        for variant in range(0, 4):
            ppt, nt, iters = (16 if variant & 1 else 8), bool(variant & 2), 1 << (variant >> 2)
            ms = timed(ctx, lambda: ctx.stream_probe(a.tiles, batch.n_pixels, batch.pin, batch.pout,
                                                     variant), a.reps)
            out[f'probe ppt={ppt} nt={int(nt)} iters={iters}'] = round(px * 21 / (sum(ms) / len(ms)) / 1e6, 1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
