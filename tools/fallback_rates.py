#!/usr/bin/env python3
"""rates of the paths that used to fall back to the slow kernels, device-resident, HIP events.
  (a) planes at ODD addresses, odd tile stride (no 16-byte alignment anywhere): the direct kernel with unaligned
      accesses (round 5: whole tiles on dswx_classify_v1, 0.169 of peak -- profiles/r05_generic_kernel_stats.csv)
  (b) 'cover' mode on a ragged contiguous batch (3660 x 3659 tiles; round 5: stage 1 on dswx_classify_v1)
  (c) terrain shadow layer with an odd margin / odd width (round 5: dswx_shadow_v2, 0.24 of peak, 2.0 x traffic), beside
      the general kernel forced on the same geometry (lab switch) and the reference's aligned geometry
Prints one JSON object (profiles/r06_fallback_rates.json)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from proteus_amd import _capi                     # noqa: E402
from proteus_amd.synth import SEED, synth_dem     # noqa: E402

PEAK = 8000.0


def timed(ctx, fn, reps=10):
    for _ in range(3 * reps):           # warm-up: the clocks ramp for the first tenths of a second after an idle period
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    ctx.synchronize()
    ms = ctx.elapsed_ms(e0, e1) / reps
    ctx.destroy_event(e0)
    ctx.destroy_event(e1)
    return ms


def rec(ms, nbytes, info=None):
    gbs = nbytes / (ms * 1e-3) / 1e9
    r = {'ms_per_launch': round(ms, 4), 'GBps_algorithmic': round(gbs, 1), 'frac': round(gbs / PEAK, 4)}
    if info:
        r['kernel'] = info
    return r


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    ctx = _capi.Context(0)
    p = _capi.default_params()
    out = {'tiles': n}
    # ---- (a) odd addresses, odd stride: classify_batch on planes carved out of one arena at odd offsets
    h = w = 3660
    P = h * w
    stride = P + 3
    arena = ctx.malloc(n * stride * 21 + 8192)
    pin, pout = _capi.PlanesIn(), _capi.PlanesOut()
    off = 2
    for k in range(6):
        pin.band[k] = arena.ptr + off
        off += n * stride * 2 + 2
    off += 1
    pin.fmask = arena.ptr + off
    off += n * stride + 1
    off += off % 2
    pout.diag = arena.ptr + off
    off += n * stride * 2 + 3
    for name in ('wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
        setattr(pout, name, arena.ptr + off)
        off += n * stride + 3
    off += (-off) % 8
    cnt = arena.ptr + off
    geom = _capi.BatchGeom(n, h, w, stride)
    ctx.synth_batch(SEED, 0, geom, pin)
    ms = timed(ctx, lambda: ctx.classify_batch(p, geom, pin, pout, cnt))
    out['a_odd_addresses_odd_stride'] = rec(ms, n * P * 21, ctx.last_kernel_info())
    out['a_odd_addresses_odd_stride']['r05_same_case'] = 'dswx_classify_v1 on whole tiles: 0.169 of peak (profiles/r05_generic_kernel_stats.csv)'
    arena.free()
    # the same bytes aligned, for scale
    b = _capi.DeviceBatch(ctx, n, h, w)
    b.synth(SEED)
    ms = timed(ctx, lambda: b.classify(p))
    out['a_aligned_for_scale'] = rec(ms, n * P * 21, ctx.last_kernel_info())
    b.free()
    # ---- (b) 'cover' on a ragged contiguous batch
    pc = _capi.make_params(mask_adjacent_to_cloud_mode='cover')
    for name, hh, ww, align in (('b_cover_ragged_3660x3659', 3660, 3659, 1), ('b_cover_padded_3660x3660_for_scale', 3660, 3660, 256)):
        b = _capi.DeviceBatch(ctx, n, hh, ww, masks=True, tile_align=align)
        b.synth(SEED)
        ms = timed(ctx, lambda: b.classify(pc), reps=5)
        out[name] = rec(ms, n * hh * ww * 24, ctx.last_kernel_info())
        b.free()
    # ---- (c) terrain shadow layer, odd geometry
    sun = [0.3, 0.4, 0.866]
    for name, side, margin in (('c_shadow_reference_geometry_3760_margin_50', 3760, 50), ('c_shadow_odd_margin_51_side_3762', 3762, 51),
                               ('c_shadow_margin_3_side_3667', 3667, 3)):
        dem = np.stack([synth_dem(t, side, side) for t in range(min(n, 4))])
        reps_t = n // dem.shape[0]
        d_dem = ctx.malloc(dem.nbytes * reps_t)
        for r in range(reps_t):
            d_dem.upload(dem.ravel(), r * dem.nbytes)
        nt = dem.shape[0] * reps_t
        oh = side - 2 * margin
        d_out = ctx.malloc(nt * oh * oh)
        nbytes = nt * (side * side * 4 + oh * oh)
        call = lambda: ctx.shadow_layer_device(d_dem.ptr, nt, side, side, margin, sun, 0.6, 0.8, -5.0, 40.0, d_out.ptr)   # noqa: E731
        out[name] = rec(timed(ctx, call), nbytes)
        ctx.lab_configure(shadow_kernel=2)
        out[name]['general_kernel_forced'] = rec(timed(ctx, call), nbytes)
        ctx.lab_configure(shadow_kernel=0)
        d_dem.free()
        d_out.free()
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == '__main__':
    main()
