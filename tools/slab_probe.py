#!/usr/bin/env python3
"""Does choosing the physical range of EVERY output plane separately pay?  (the fused kernel's
rate follows the write side, and the write rate is a stable property of the physical range.)

The inputs of a T-tile batch get one allocation.  Then as many plane-sized slabs as fit are allocated side by side
(u8 planes: T x stride bytes; DIAG: twice that), each slab's write rate is measured with a device fill, and the
fused kernel is timed with its seven output planes bound to
  first   the first slabs handed out (what a plain allocation would give),
  best    the slabs with the highest write rate,
  worst   the slabs with the lowest.

    python tools/slab_probe.py [--tiles 256] [--reps 5]
"""
import argparse
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteus_amd import _capi            # noqa: E402
from proteus_amd.synth import SEED       # noqa: E402

T = 3660
U8 = ('wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud')


class Hip:
    def __init__(self):
        self.lib = ctypes.CDLL('libamdhip64.so')

    def check(self, rc):
        if rc:
            raise RuntimeError(f'hip error {rc}')

    def fill_rate(self, ptr, nbytes, reps=3):
        """GB/s of hipMemsetAsync over [ptr, ptr + nbytes) on the null stream."""
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        self.check(self.lib.hipEventCreate(ctypes.byref(e0)))
        self.check(self.lib.hipEventCreate(ctypes.byref(e1)))
        self.check(self.lib.hipMemsetAsync(ctypes.c_void_p(ptr), 0, ctypes.c_size_t(nbytes), None))
        self.check(self.lib.hipEventRecord(e0, None))
        for _ in range(reps):
            self.check(self.lib.hipMemsetAsync(ctypes.c_void_p(ptr), 0, ctypes.c_size_t(nbytes), None))
        self.check(self.lib.hipEventRecord(e1, None))
        self.check(self.lib.hipEventSynchronize(e1))
        ms = ctypes.c_float()
        self.check(self.lib.hipEventElapsedTime(ctypes.byref(ms), e0, e1))
        self.lib.hipEventDestroy(e0)
        self.lib.hipEventDestroy(e1)
        return nbytes * reps / ms.value / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tiles', type=int, default=256)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--diag-slabs', type=int, default=8)
    ap.add_argument('--max-slabs', type=int, default=48)
    ap.add_argument('--keep-free-gb', type=float, default=10.0)
    a = ap.parse_args()
    import torch
    ctx = _capi.Context(0)
    hip = Hip()
    params = _capi.default_params()
    b = _capi.DeviceBatch(ctx, a.tiles, T, T, separate_outputs=True)
    b.synth(SEED)
    ctx.synchronize()
    for buf in b.out_bufs.values():     # the batch's own output planes are not used here
        buf.free()
    b.out_bufs = {}
    S = a.tiles * b.tile_stride
    free_bytes, _ = torch.cuda.mem_get_info()
    budget = free_bytes - int(a.keep_free_gb * 1e9)
    big = [ctx.malloc(2 * S) for _ in range(a.diag_slabs)]
    budget -= 2 * S * a.diag_slabs
    small = [ctx.malloc(S) for _ in range(max(6, min(a.max_slabs, budget // S)))]
    rate_big = [hip.fill_rate(x.ptr, 2 * S) for x in big]
    rate_small = [hip.fill_rate(x.ptr, S) for x in small]
    rate_small2 = [hip.fill_rate(x.ptr, S) for x in small]          # stability of the per-slab figure

    def bind(diag_buf, u8_bufs):
        b.pout.diag = diag_buf.ptr
        for name, buf in zip(U8, u8_bufs):
            setattr(b.pout, name, buf.ptr)

    def rate():
        for _ in range(2):
            b.classify(params)
        ctx.synchronize()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(a.reps):
            b.classify(params)
        ctx.record(e1)
        ctx.synchronize()
        ms = ctx.elapsed_ms(e0, e1) / a.reps
        ctx.destroy_event(e0)
        ctx.destroy_event(e1)
        return round(a.tiles * T * T * 21 / ms / 1e6, 1)

    import random
    rng = random.Random(7)
    order_b = sorted(range(len(big)), key=lambda i: -rate_big[i])
    order_s = sorted(range(len(small)), key=lambda i: -rate_small[i])
    out = {'tiles': a.tiles, 'slab_GB': round(S / 1e9, 3), 'n_small': len(small), 'n_big': len(big),
           'fill_GBps_small_minmax': [round(min(rate_small)), round(max(rate_small))],
           'fill_GBps_big_minmax': [round(min(rate_big)), round(max(rate_big))], 'fused_GBps': {}}
    sets = [('first', 0, list(range(6))), ('best_fill', order_b[0], order_s[:6]), ('worst_fill', order_b[-1], order_s[-6:])]
    for k in range(10):
        sets.append((f'random{k}', rng.randrange(len(big)), rng.sample(range(len(small)), 6)))
    for rnd in range(2):                       # interleaved, two rounds
        for name, d, u in sets:
            bind(big[d], [small[i] for i in u])
            out['fused_GBps'].setdefault(name, []).append(rate())
    out['sets'] = {name: [d] + u for name, d, u in sets}
    # one plane at a time over every slab, the others fixed (twice: is a slab's figure repeatable?)
    base_d, base_u = sets[3][1], sets[3][2]
    for rnd in range(2):
        row = []
        for d in range(len(big)):
            bind(big[d], [small[i] for i in base_u])
            row.append(rate())
        out.setdefault('diag_sweep', []).append(row)
    free_small = [i for i in range(len(small)) if i not in base_u[:5]]
    for rnd in range(2):
        row = []
        for i in free_small:
            bind(big[base_d], [small[j] for j in base_u[:5]] + [small[i]])
            row.append(rate())
        out.setdefault('cloud_sweep', []).append(row)
    out['cloud_sweep_slabs'] = free_small
    # and the READ side: band 0 copied into every DIAG-sized slab in turn (DIAG stays where the sets above put it)
    bind(big[base_d], [small[i] for i in base_u])
    band0 = b.pin.band[0]
    for rnd in range(2):
        row = []
        for d in range(len(big)):
            if d == base_d:
                row.append(None)
                continue
            hip.check(hip.lib.hipMemcpy(ctypes.c_void_p(big[d].ptr), ctypes.c_void_p(band0), ctypes.c_size_t(2 * S), 3))
            b.pin.band[0] = big[d].ptr
            row.append(rate())
        b.pin.band[0] = band0
        row.append(rate())             # last entry: band 0 where the arena has it
        out.setdefault('band0_sweep', []).append(row)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
