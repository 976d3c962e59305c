#!/usr/bin/env python3
"""ROUND 6 NOTE: the direct kernel takes ragged batches itself now (unaligned accesses), so `fused_variant=0` below no longer
falls through to the generic kernel: the entries named generic_* of a run of THIS version measure dswx_classify_v8 on the
ragged batch (see the `kernel` field; tools/fallback_rates.py is the round-6 harness; profiles/r05_generic_kernel_rate.json
keeps the generic kernel's 0.15 - 0.17).  Original description:

Rate of the one-pixel-per-thread generic kernel `dswx_classify_v1` (VERDICT r04 "What's missing" 2: no rate for it
existed anywhere) on a RAGGED contiguous batch -- 3660 x 3659 tiles: H*W = 4 (mod 8), so tiles 1.. start off the 8-byte
grid of the u8 planes -- which until round 5 was the only kernel that could run such a batch, and of the table-driven
kernel on the same batch since it learnt to start every tile at its first 8-pixel boundary (KArgs::ragged; the generic
kernel then only does the < 8 + < 8 edge pixels of each tile).  The generic kernel is forced through the lab switch
`fused_variant=0` (the direct kernel cannot take a ragged batch, so the dispatch falls through to it).  Integer and
float32 chain; for comparison in the same process the vector kernels on 3660 x 3660 tiles of the same count.  Prints one
JSON object; run under `rocprofv3 --kernel-trace --stats` for the trace (profiles/r05_generic_kernel_stats.csv)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from proteus_amd import _capi                     # noqa: E402
from proteus_amd.synth import SEED                # noqa: E402
import bench                                      # noqa: E402  (the --scaled parameter set)

PEAK = 8000.0


def rate(ctx, batch, params, reps):
    batch.classify(params)
    ctx.synchronize()
    info = ctx.last_kernel_info()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        batch.classify(params)
    ctx.record(e1)
    ctx.synchronize()
    ms = ctx.elapsed_ms(e0, e1) / reps
    ctx.destroy_event(e0)
    ctx.destroy_event(e1)
    gbs = batch.n_tiles * batch.n_pixels * 21 / (ms * 1e-3) / 1e9
    return {'kernel': info, 'ms_per_launch': round(ms, 4), 'GBps_of_21_B_per_px': round(gbs, 1), 'frac': round(gbs / PEAK, 4)}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ctx = _capi.Context(0)
    import argparse
    p_int = _capi.default_params()
    p_f32 = bench.bench_params(argparse.Namespace(scaled=True))
    out = {'tiles': n}
    ragged = _capi.DeviceBatch(ctx, n, 3660, 3659, tile_align=1)       # 13,391,940 px per tile: 4 (mod 8)
    ragged.synth(SEED)
    out['ragged_vector_int16_f64'] = rate(ctx, ragged, p_int, reps)
    out['ragged_vector_float32'] = rate(ctx, ragged, p_f32, reps)
    ragged.free()
    forced = _capi.Context(0)                                          # its own context: the switch is per context
    forced.lab_configure(fused_variant=0)
    ragged = _capi.DeviceBatch(forced, n, 3660, 3659, tile_align=1)
    ragged.synth(SEED)
    out['generic_int16_f64'] = rate(forced, ragged, p_int, reps)
    out['generic_float32'] = rate(forced, ragged, p_f32, reps)
    ragged.free()
    forced.close()
    for name, align in (('padded', 256), ('contiguous', 1)):
        b = _capi.DeviceBatch(ctx, n, 3660, 3660, tile_align=align)
        b.synth(SEED)
        out[f'vector_int16_f64_{name}'] = rate(ctx, b, p_int, reps)
        out[f'vector_float32_{name}'] = rate(ctx, b, p_f32, reps)
        b.free()
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == '__main__':
    main()
