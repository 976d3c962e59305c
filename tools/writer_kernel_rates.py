#!/usr/bin/env python3
"""Rates of the reader / writer kernels of ABI v6 (csrc/dswx_writer.hip), device-resident, HIP events, warm clocks: bytes
read + written per launch over the launch time, as a fraction of the 8 TB/s HBM peak.  In a product run these kernels sit
between two PCIe copies of the same bytes (~55 GB/s) and a DEFLATE codec: their rate is never what a product waits for;
the numbers are here so that every kernel of the library has one.  Prints one JSON object (profiles/r06_writer_kernel_rates.json)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from proteus_amd import _capi, geotiff            # noqa: E402

PEAK = 8000.0
H = W = 3660
FACTORS = geotiff.COG_OVERVIEW_FACTORS


def timed(ctx, fn, reps=5, inner=20):
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
    return sum(ms) / len(ms)


def rec(ms, nbytes, note=None):
    gbs = nbytes / (ms * 1e-3) / 1e9
    r = {'us_per_launch': round(ms * 1e3, 1), 'bytes_per_launch': int(nbytes), 'GBps': round(gbs, 1), 'frac_of_8TBps': round(gbs / PEAK, 4)}
    if note:
        r['note'] = note
    return r


def main():
    ctx = _capi.Context(0)
    rng = np.random.default_rng(1)
    out = {'raster': [H, W], 'tile': 512}
    for name, dtype in (('u8', np.uint8), ('u16', np.uint16)):
        es = np.dtype(dtype).itemsize
        arr = rng.integers(0, 5, size=(H, W)).astype(dtype)
        lay = _capi.cog_layout(H, W, es, FACTORS)
        d_plane, d_blocks = ctx.malloc(arr.nbytes), ctx.malloc(lay['total_bytes'])
        d_plane.upload(arr)
        ms = timed(ctx, lambda: ctx.cog_blocks_device(d_plane.ptr, es, H, W, d_blocks.ptr, FACTORS, 512, 2))
        out[f'dswx_cog_blocks_v1<{name}> image + 4 NEAREST overviews, PREDICTOR=2'] = rec(
            ms, arr.nbytes + lay['total_bytes'], 'reads the plane once (+ 6.7 % gathered for the overviews), writes the padded blocks (1.39 x the plane)')
        lay0 = _capi.cog_layout(H, W, es, ())
        ms = timed(ctx, lambda: ctx.untile_device(d_blocks.ptr, es, H, W, 512, 512, 2, d_plane.ptr))
        out[f'dswx_untile_v1<{name}> 512^2 tiles, PREDICTOR=2'] = rec(ms, arr.nbytes + lay0['total_bytes'])
        d_plane.free()
        d_blocks.free()
    f = rng.normal(0.1, 0.05, size=(H, W)).astype(np.float32)
    lay = _capi.cog_layout(H, W, 4, ())
    d_plane, d_blocks = ctx.malloc(f.nbytes), ctx.malloc(lay['total_bytes'])
    d_plane.upload(f)
    ms = timed(ctx, lambda: ctx.cog_blocks_device(d_plane.ptr, 4, H, W, d_blocks.ptr, (), 512, 3))
    out['dswx_cog_blocks_f32 (floating-point predictor)'] = rec(ms, f.nbytes + lay['total_bytes'])
    ms = timed(ctx, lambda: ctx.untile_device(d_blocks.ptr, 4, H, W, 512, 512, 3, d_plane.ptr))
    out['dswx_untile (Float32, PREDICTOR=3: byte running sums + byte-plane gather)'] = rec(
        ms, f.nbytes + 3 * lay['total_bytes'], 'two launches; the scratch of the running sums is written and read once more')
    d_blocks.free()
    bands = [ctx.malloc(H * W * 2) for _ in range(3)]
    diag = ctx.malloc(H * W * 2)
    rgb = ctx.malloc(3 * H * W * 4)
    for b in bands + [diag]:
        b.upload(rng.integers(0, 9000, size=H * W).astype(np.uint16))
    ms = timed(ctx, lambda: ctx.rgb_planes_device(bands[0].ptr, bands[1].ptr, bands[2].ptr, diag.ptr, H * W, [1e-4] * 3, [0.0] * 3, True, rgb.ptr))
    out['dswx_rgb_planes_v1'] = rec(ms, H * W * (8 + 12))
    side = H + 100
    dem = ctx.malloc(side * side * 4)
    ms = timed(ctx, lambda: ctx.copy_2d_device(d_plane.ptr, W * 4, dem.ptr + (50 * side + 50) * 4, side * 4, W * 4, H))
    out['dswx_copy_2d_device (DEM crop, hipMemcpy2DAsync)'] = rec(ms, 2 * H * W * 4)
    # CUBICSPLINE level 4 of a Float32 layer: horizontal pass (float32 -> float64 [H][ow]) and vertical pass (-> float32 [oh][ow])
    oh = ow = -(-H // 4)
    fx, wx = geotiff.convolve_weights(W, ow)
    d_first, d_w = ctx.malloc(fx.size * 4), ctx.malloc(wx.nbytes)
    d_first.upload(fx.astype(np.int32))
    d_w.upload(np.ascontiguousarray(wx.T))
    tmp, lvl = ctx.malloc(H * ow * 8), ctx.malloc(oh * ow * 4)
    taps = wx.shape[1]
    ms = timed(ctx, lambda: ctx.convolve_axis_device(d_plane.ptr, False, H, W, W, 1, ow, taps, d_first.ptr, d_w.ptr, tmp.ptr, True, ow, 1))
    out['dswx_convolve_axis_v1<float, double> (CUBICSPLINE level 4, horizontal pass, 17 taps)'] = rec(
        ms, H * W * 4 + H * ow * 8, 'every output reads 17 taps: 4.25 x the plane through the caches, once from memory')
    ms = timed(ctx, lambda: ctx.convolve_axis_device(tmp.ptr, True, ow, H, 1, ow, oh, taps, d_first.ptr, d_w.ptr, lvl.ptr, False, 1, ow))
    out['dswx_convolve_axis_v1<double, float> (CUBICSPLINE level 4, vertical pass, 17 taps)'] = rec(ms, H * ow * 8 + oh * ow * 4)
    # Byte conversion of the multi-band file's DIAG / DEM bands; the browse gather 3660^2 -> 1024^2
    byte = ctx.malloc(H * W)
    ms = timed(ctx, lambda: ctx.to_byte_device(d_plane.ptr, np.float32, H * W, byte.ptr))
    out['dswx_to_byte_v1<float>'] = rec(ms, H * W * 5)
    ms = timed(ctx, lambda: ctx.to_byte_device(diag.ptr, np.uint16, H * W, byte.ptr))
    out['dswx_to_byte_v1<uint16>'] = rec(ms, H * W * 3)
    ys, xs = geotiff.resample_nearest_indices(H, W, 1024, 1024)
    d_idx = ctx.malloc(2048 * 4)
    d_idx.upload(np.concatenate([ys, xs]).astype(np.int32))
    ms = timed(ctx, lambda: ctx.gather_2d_device(byte.ptr, 1, H, W, d_idx.ptr, 1024, d_idx.ptr + 4096, 1024, lvl.ptr))
    out['dswx_gather_2d_v1<u8> (browse 3660^2 -> 1024^2)'] = rec(ms, 2 * 1024 * 1024, 'bytes: one read and one write per OUTPUT pixel (a latency figure)')
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == '__main__':
    main()
