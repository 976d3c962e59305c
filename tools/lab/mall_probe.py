#!/usr/bin/env python3
"""which part of the stencil kernels' over-fetch reaches HBM?  FETCH_SIZE / WRITE_SIZE (and
TCC_EA0_RDREQ_DRAM) are counted at the L2's fabric side: an Infinity Cache hit looks like an HBM read.  No counter of this
stack sits behind the Infinity Cache, so the split is taken by the guide's own method -- reuse distance against the
256 MiB of the cache:

  shadow   the filter kernel re-reads 2 halo rows per 8 output rows (fabric traffic 1.27 x algorithmic).  In ONE launch the
           two reads of a halo row are a few hundred KiB of traffic apart; launched in TWO passes (lab switch
           shadow_kernel=3: even block rows of every tile, then the odd ones) they are a whole pass apart -- 1 GB at 32
           tiles, far beyond the cache; ~140 MB at 4 tiles, inside it.  Same instructions, same fabric traffic
           (a FETCH_SIZE pass of this script shows it); the time difference at 32 tiles that is absent at 4 tiles is what
           the Infinity Cache absorbs in the one-launch order.
  cover    the state byte + bitmaps are written by stage 1 and read back by stages 2 / 3; the distance is the whole
           batch: per-tile time against the tile count shows where the round trip leaves the cache.

Prints one JSON object.  `--pmc-run` does a short fixed sequence for a rocprofv3 --pmc pass."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from proteus_amd import _capi                     # noqa: E402
from proteus_amd.synth import SEED, synth_dem     # noqa: E402


def timed(ctx, fn, reps=5, inner=10):
    for _ in range(3 * inner):
        fn()
    ctx.synchronize()
    ms = []
    for _ in range(reps):
        a, b = ctx.event(), ctx.event()
        ctx.record(a)
        for _ in range(inner):
            fn()
        ctx.record(b)
        ctx.synchronize()
        ms.append(ctx.elapsed_ms(a, b) / inner)
        ctx.destroy_event(a)
        ctx.destroy_event(b)
    return sum(ms) / len(ms), min(ms)


def main():
    pmc_run = '--pmc-run' in sys.argv
    ctx = _capi.Context(0)
    side, margin = 3760, 50
    oh = side - 2 * margin
    sun = [0.3, 0.4, 0.866]
    out = {}
    base = np.stack([synth_dem(t, side, side) for t in range(4)])
    for nt in (32, 4):
        d_dem = ctx.malloc(base.nbytes * (nt // 4))
        for r in range(nt // 4):
            d_dem.upload(base.ravel(), r * base.nbytes)
        d_out = ctx.malloc(nt * oh * oh)
        call = lambda: ctx.shadow_layer_device(d_dem.ptr, nt, side, side, margin, sun, 0.6, 0.8, -5.0, 40.0, d_out.ptr)   # noqa: E731
        algorithmic = nt * (side * side * 4 + oh * oh)
        rec = {'tiles': nt, 'dem_MB': round(nt * side * side * 4 / 1e6, 1), 'algorithmic_MB': round(algorithmic / 1e6, 1)}
        if pmc_run:
            for mode in (0, 3):
                ctx.lab_configure(shadow_kernel=mode)
                for _ in range(3):
                    call()
                ctx.synchronize()
            ctx.lab_configure(shadow_kernel=0)
        else:
            for name, mode in (('one_launch', 0), ('two_passes_even_odd_block_rows', 3), ('one_launch_again', 0)):
                ctx.lab_configure(shadow_kernel=mode)
                avg, best = timed(ctx, call)
                rec[name] = {'ms': round(avg, 4), 'ms_min': round(best, 4), 'algorithmic_GBps': round(algorithmic / avg / 1e6, 1)}
            ctx.lab_configure(shadow_kernel=0)
            rec['two_passes_over_one_launch'] = round(rec['two_passes_even_odd_block_rows']['ms'] /
                                                      (0.5 * (rec['one_launch']['ms'] + rec['one_launch_again']['ms'])), 4)
        out[f'shadow_{nt}_tiles'] = rec
        d_dem.free()
        d_out.free()
    if not pmc_run:
        pc = _capi.make_params(mask_adjacent_to_cloud_mode='cover')
        pm = _capi.default_params()
        cov = {}
        for nt in (1, 2, 4, 8, 16, 32):
            b = _capi.DeviceBatch(ctx, nt, 3660, 3660, masks=True)
            b.synth(SEED)
            avg_c, _ = timed(ctx, lambda: b.classify(pc), reps=3, inner=5)
            avg_m, _ = timed(ctx, lambda: b.classify(pm), reps=3, inner=5)
            cov[str(nt)] = {'cover_ms_per_tile': round(avg_c / nt, 5), 'mask_mode_ms_per_tile': round(avg_m / nt, 5),
                            'cover_minus_mask_ms_per_tile': round((avg_c - avg_m) / nt, 5),
                            'planes_MB': round(nt * 13395600 * 24 / 1e6), 'cover_scratch_MB': round(nt * 13395600 * 1.16 / 1e6)}
            b.free()
        out['cover_per_tile_vs_tile_count'] = cov
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == '__main__':
    main()
