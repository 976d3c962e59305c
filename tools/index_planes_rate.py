#!/usr/bin/env python3
"""Rate of `dswx_indices_v1`, the kernel behind north_star's "float spectral indices": float64 MNDWI / NDVI / AWESH planes
(true IEEE division, :1872-1887) of a device-resident batch -- 12 B read (six int16 bands) + 24 B written per pixel.
A debug / validation output, not on the product's default path (the classifier never materialises the indices); measured
here so that every kernel of the library has a number.  Times dswx_classify_batch with and without the three planes
requested (HIP events); run under `rocprofv3 --kernel-trace --stats` for the kernel's own duration.  One JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from proteus_amd import _capi                     # noqa: E402
from proteus_amd.synth import SEED                # noqa: E402


def timed(ctx, fn, reps):
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ctx = _capi.Context(0)
    b = _capi.DeviceBatch(ctx, n, 3660, 3660)
    b.synth(SEED)
    p = _capi.default_params()
    px = n * b.tile_stride
    planes = [ctx.malloc(px * 8) for _ in range(3)]
    pout = _capi.PlanesOut()
    for name in ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
        setattr(pout, name, getattr(b.pout, name))
    plain = timed(ctx, lambda: ctx.classify_batch(p, b.geom, b.pin, pout, b.counters_ptr), reps)
    pout.mndwi, pout.ndvi, pout.awesh = (x.ptr for x in planes)
    both = timed(ctx, lambda: ctx.classify_batch(p, b.geom, b.pin, pout, b.counters_ptr), reps)
    ms = both - plain
    bytes_alg = n * 3660 * 3660 * 36
    print(json.dumps({'tiles': n, 'classifier_ms': round(plain, 4), 'classifier_plus_index_planes_ms': round(both, 4),
                      'dswx_indices_v1_ms': round(ms, 4), 'algorithmic_bytes': bytes_alg,
                      'GBps': round(bytes_alg / ms / 1e6, 1), 'frac_of_8TBps': round(bytes_alg / ms / 1e6 / 8000, 4),
                      'kernel': ctx.last_kernel_info()}, indent=1))
    for x in planes:
        x.free()
    b.free()
    ctx.close()


if __name__ == '__main__':
    main()
