#!/usr/bin/env python3
"""Where does the allocation dependence of the fused kernel's rate come from?  

One large allocation; the 14 planes of a T-tile batch are laid out inside it at a configurable base offset and
with a configurable extra gap between consecutive planes, the batch is generated and classified, the rate
measured.  If the rate moves with base / gap inside ONE allocation, the placement of the streams relative to each
other (or to the part's channel / bank hashing) matters and a layout rule could fix it; if it only moves between
allocations, it is the physical pages the driver handed out.

    python tools/lab/placement_probe.py [--tiles 64] [--reps 5]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from proteus_amd import _capi            # noqa: E402
from proteus_amd.synth import SEED       # noqa: E402

T = 3660


def layout(base_ptr, base, gap, n_tiles, stride):
    pin, pout = _capi.PlanesIn(), _capi.PlanesOut()
    off = base
    total = n_tiles * stride

    def take(nbytes):
        nonlocal off
        p = base_ptr + off
        off += ((nbytes + 255) & ~255) + gap
        return p
    for i in range(6):
        pin.band[i] = take(total * 2)
    pin.fmask = take(total)
    pout.diag = take(total * 2)
    for name in ('wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
        setattr(pout, name, take(total))
    counters = take(n_tiles * 24)
    return pin, pout, counters, off


def rate(ctx, params, geom, pin, pout, counters, reps, px):
    for _ in range(3):
        ctx.classify_batch(params, geom, pin, pout, counters)
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        ctx.classify_batch(params, geom, pin, pout, counters)
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1) / reps
    ctx.destroy_event(e0)
    ctx.destroy_event(e1)
    return px * 21 / ms / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tiles', type=int, default=64)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--arena-gb', type=float, default=60.0)
    a = ap.parse_args()
    ctx = _capi.Context(0)
    params = _capi.default_params()
    stride = -(-T * T // 256) * 256
    geom = _capi.BatchGeom(a.tiles, T, T, stride)
    px = a.tiles * T * T
    out = {'tiles': a.tiles, 'allocations': []}
    MB, GB = 1 << 20, 1 << 30
    cases = [(0, 0), (0, 4096), (0, 2 * MB), (0, 2 * MB + 4096), (0, 64 * MB), (0, 256 * MB + 8192), (0, GB),
             (GB, 0), (5 * GB + 2 * MB, 0), (13 * GB, 0), (0, 0)]
    for alloc in range(3):
        arena = ctx.malloc(int(a.arena_gb * GB))
        rows = []
        for base, gap in cases:
            pin, pout, counters, end = layout(arena.ptr, base, gap, a.tiles, stride)
            if end > arena.nbytes:
                continue
            ctx.synth_batch(SEED, 0, geom, pin)
            rows.append({'base_MB': base // MB, 'gap_KB': gap // 1024,
                         'GBps': round(rate(ctx, params, geom, pin, pout, counters, a.reps, px), 1)})
        out['allocations'].append(rows)
        arena.free()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
