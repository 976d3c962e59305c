#!/usr/bin/env python3
"""Measurement of the SURVEY 8(f) rows next to the hot path, device-resident, HIP events:
terrain shadow layer (f1), 'cover' mode (f2), LAND 3x3 aggregation (f3).  For each: time per
3660^2 tile, achieved GB/s of the algorithmic bytes, and the numpy oracle on one host core on a
bounded sample.  Prints one JSON object (profiles/rNN_next_rows.json).

    python tools/next_rows_bench.py [--tiles 8] [--reps 5] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteus_amd import _capi                                   # noqa: E402
from proteus_amd.synth import SEED, synth_dem, synth_landcover_inputs, synth_tile   # noqa: E402

T, MARGIN = 3660, 50


def timed(ctx, fn, reps, inner=10):
    """Average over `reps` timings of `inner` back-to-back launches each (HIP events on the kernel's stream):
    a single ~0.1 ms launch between two events also measures ~10 us of dispatch gap."""
    for _ in range(3 * inner):          # warm-up: clocks ramp for the first tenths of a second after an idle period
        fn()
    ctx.synchronize()
    ms = []
    for _ in range(reps):
        a, b = ctx.event(), ctx.event()
        ctx.record(a)
        for _ in range(inner):
            fn()
        ctx.record(b)
        ms.append(ctx.elapsed_ms(a, b) / inner)
        ctx.destroy_event(a)
        ctx.destroy_event(b)
    return sum(ms) / len(ms), min(ms)


def coherent_fmask(fmask, seed, clear_noise=True):
    """tests/test_gpu_parity.py::blobby_fmask at full tile size: the same patches (adjacent-to-cloud diamonds, snow
    discs), stamped into local windows instead of evaluated over the whole raster per patch.  clear_noise: the
    white-noise snow bits of the synthetic Fmask are dropped as well, so that snow exists in the discs only
    (False = exactly the test generator)."""
    h, w = fmask.shape
    rng = np.random.default_rng(seed)
    adj = np.zeros((h, w), bool)
    snow = np.zeros((h, w), bool)

    def stamp(dst, cy, cx, ry, rx, inside):
        y0, y1, x0, x1 = max(cy - ry, 0), min(cy + ry + 1, h), max(cx - rx, 0), min(cx + rx + 1, w)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        dst[y0:y1, x0:x1] |= inside(yy - cy, xx - cx)
    for _ in range(max(3, h * w // 1500)):
        cy, cx, r = int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(2, 14))
        stamp(adj, cy, cx, r, 2 * r + 1, lambda dy, dx: (np.abs(dy) + np.abs(dx) // 2) < r)
        cy, cx, r = int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(1, 5))
        stamp(snow, cy, cx, r, r, lambda dy, dx: dy * dy + dx * dx < r * r)
    valid = fmask != 255
    out = np.where(valid & adj, (fmask | 4) & ~np.uint8(2 | 8), fmask & ~np.uint8(4)).astype(np.uint8)
    if clear_noise:
        out = np.where(valid, out & ~np.uint8(16), out).astype(np.uint8)
    return np.where(valid & snow, out | 16, out).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tiles', type=int, default=32,
                    help='tiles per launch (8 tiles = 0.1 ms launches whose ramp / tail cost ~10 %%)')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--unplaced', action='store_true', help="'cover' batch in one plain allocation (no placement)")
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--lab', action='store_true', help='also time lab A/B switches (libdswx_lab.so)')
    a = ap.parse_args()
    n = a.tiles
    ctx = _capi.Context(0)
    out = {'tiles_per_launch': n, 'tile': [T, T]}

    # ---- f1 terrain shadow: [n][3760][3760] float32 -> [n][3660][3660] u8
    az, el = np.radians(141.0), 35.0
    zen = np.radians(90 - el)
    sun = [np.sin(az) * np.sin(zen), np.cos(az) * np.sin(zen), np.cos(zen)]
    out['f1_shadow'] = {}
    # margin 50 (the reference's) -> dswx_shadow_v3 (four pixels per thread behind the filter);
    # margin 51 on a 3762^2 DEM -> the general one-pixel kernel dswx_shadow_v2 on the same 3660^2 outputs
    for tag, margin in (('quads_filter_v3', MARGIN), ('general_v2', MARGIN + 1)):
        H = W = T + 2 * margin
        dem = synth_dem(3, H, W)
        d_dem = ctx.malloc(n * dem.nbytes)
        d_sh = ctx.malloc(n * T * T)
        for t in range(n):
            d_dem.upload(dem, t * dem.nbytes)
        nbytes = n * (H * W * 4 + T * T)
        for mode, f32 in (('nep50', False), ('legacy_f32', True)):
            avg, mn = timed(ctx, lambda: ctx.shadow_layer_device(d_dem.ptr, n, H, W, margin, sun, np.sin(az), np.cos(az),
                                                                 -5.0, 40.0, d_sh.ptr, float32=f32), a.reps)
            out['f1_shadow'][f'{tag}_{mode}'] = {
                'ms_per_tile': avg / n, 'ms_min_per_tile': mn / n, 'algorithmic_bytes_per_tile': nbytes // n,
                'GBps': nbytes / avg / 1e6, 'frac_of_8TBps': nbytes / avg / 1e6 / 8000, 'Mpix_s': n * T * T / avg / 1e3}
        if a.lab and tag == 'quads_filter_v3':
            # launch-geometry A/B through the lab switch: grid.x as it comes (15 for 3660 columns) vs padded to 8 | grid.x
            for pad in (1, 8):
                c2 = _capi.Context(0)
                c2.lab_configure(shadow_grid_pad=pad)
                avg, mn = timed(c2, lambda: c2.shadow_layer_device(d_dem.ptr, n, H, W, margin, sun, np.sin(az), np.cos(az),
                                                                   -5.0, 40.0, d_sh.ptr, float32=True), a.reps)
                out['f1_shadow'][f'ab_grid_pad_{pad}_legacy_f32'] = {'ms_per_tile': avg / n, 'ms_min_per_tile': mn / n}
                c2.close()
        d_dem.free()
        d_sh.free()
    dem = synth_dem(3, T + 2 * MARGIN, T + 2 * MARGIN)

    # ---- f3 LAND aggregation: [n][10980][10980] u8 + [n][3660][3660] u8 -> [n][3660][3660] u8
    wc, cg = synth_landcover_inputs(2, T, T)
    d_wc, d_cg, d_land = ctx.malloc(n * wc.nbytes), ctx.malloc(n * cg.nbytes), ctx.malloc(n * T * T)
    for t in range(n):
        d_wc.upload(wc, t * wc.nbytes)
        d_cg.upload(cg, t * cg.nbytes)
    forest = [111, 113, 115, 116, 121, 123, 125, 126]
    avg, mn = timed(ctx, lambda: ctx.landcover_mask_device(d_wc.ptr, d_cg.ptr, n, T, T, forest, d_land.ptr), a.reps)
    nbytes = n * T * T * 11
    out['f3_landcover'] = {'kernel': 'dswx_landcover', 'ms_per_tile': avg / n, 'ms_min_per_tile': mn / n,
                           'algorithmic_bytes_per_tile': T * T * 11, 'GBps': nbytes / avg / 1e6,
                           'frac_of_8TBps': nbytes / avg / 1e6 / 8000,
                           'Mpix_s': n * T * T / avg / 1e3}
    for b in (d_wc, d_cg, d_land):
        b.free()

    # ---- f2 'cover' mode: the split path on a device batch with masks
    pc = _capi.make_params(mask_adjacent_to_cloud_mode='cover')
    pm = _capi.make_params(mask_adjacent_to_cloud_mode='mask')
    # the output planes placed like the headline batch's (dswx_batch_place_slide, the 'cover' launch itself as the
    # probe): unplaced, the same code measured 0.0666 and 0.0732 ms per tile in two rounds -- allocation luck
    batch = _capi.DeviceBatch(ctx, n, T, T, masks=True, sliding_outputs=not a.unplaced)
    batch.synth(SEED)
    ctx.synchronize()
    placement = None if a.unplaced else batch.place_slide(pc, slack_bytes=24 << 30, step_bytes=1 << 30)
    avg, mn = timed(ctx, lambda: batch.classify(pc), a.reps)
    info = ctx.last_kernel_info()
    avg_m, _ = timed(ctx, lambda: batch.classify(pm), a.reps)
    out['f2_cover_mode'] = {'kernel': info, 'output_plane_placement': placement, 'ms_per_tile': avg / n, 'ms_min_per_tile': mn / n,
                            'fused_mask_mode_ms_per_tile': avg_m / n,
                            'algorithmic_bytes_per_tile': T * T * 24, 'GBps_of_24B_per_px': n * T * T * 24 / avg / 1e6,
                            'frac_of_8TBps': n * T * T * 24 / avg / 1e6 / 8000,
                            'Mpix_s': n * T * T / avg / 1e3}
    # the same on spatially COHERENT Fmask: the synthetic Fmask is white noise, the worst case
    # for the dilations' early exit (every window holds snow and adjacent pixels); real Fmask has patches
    fm = coherent_fmask(batch.read_tile('fmask', 0), 1234)
    for t in range(n):
        batch.write_tile('fmask', t, fm)
    avg_c, mn_c = timed(ctx, lambda: batch.classify(pc), a.reps)
    avg_cm, _ = timed(ctx, lambda: batch.classify(pm), a.reps)
    out['f2_cover_mode']['coherent_fmask'] = {
        'ms_per_tile': avg_c / n, 'ms_min_per_tile': mn_c / n, 'fused_mask_mode_ms_per_tile': avg_cm / n,
        'frac_of_8TBps': n * T * T * 24 / avg_c / 1e6 / 8000,
        'fmask': 'adjacent-to-cloud diamonds (radius 2-13) and snow discs (radius 1-4), one per 1500 pixels, stamped '
                 'on tile 0\'s synthetic Fmask with its white-noise snow and adjacent bits cleared; the same plane in every tile',
        'pixels_adjacent': float((fm[fm != 255] & 4 != 0).mean()), 'pixels_snow': float((fm[fm != 255] & 16 != 0).mean())}
    if a.lab:
        # window width A/B of the stage-2 kernel through the lab switch (libdswx_lab.so)
        for name, switch in (('window_4_words', {'cover_kernel': 4}), ('window_8_words_direct', {'cover_kernel': 24}),
                             ('stage1_3_waves_per_simd', {'tune_lut_wps': 3})):
            c4 = _capi.Context(0)
            c4.lab_configure(**switch)
            avg4, mn4 = timed(c4, lambda: c4.classify_batch(pc, batch.geom, batch.pin, batch.pout, batch.counters_ptr), a.reps)
            out['f2_cover_mode']['ab_' + name] = {'kernel': c4.last_kernel_info(), 'ms_per_tile': avg4 / n,
                                                  'ms_min_per_tile': mn4 / n}
            c4.close()
    batch.free()

    if not a.no_cpu:
        # the oracle is only reachable through bench.py's CPU-baseline leg
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        cpu = bench.cpu_baseline_next_rows(dem, wc, cg, forest)
        out['f1_shadow']['cpu_oracle_s_per_tile'] = cpu['shadow_s_per_tile']
        out['f3_landcover']['cpu_oracle_s_per_tile'] = cpu['landcover_s_per_tile']
        out['f2_cover_mode']['cpu_oracle_s_per_tile'] = cpu['cover_s_per_tile']
        out['cpu_note'] = cpu['note']
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
