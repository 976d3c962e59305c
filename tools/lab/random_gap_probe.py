#!/usr/bin/env python3
"""Can the gain of scattered output planes (round 2's slab probe, profiles/r02_slab_probe.json) be had inside ONE allocation?

One big allocation; the 14 planes of a T-tile batch are laid out with per-plane RANDOM gaps (multiples of `--quantum`)
instead of the uniform gaps tools/lab/placement_probe.py tried.  If random in-arena layouts reach what separately
allocated planes reach, the effect is the relative position of the streams and a layout rule can buy it; if
they stay at the packed rate, it is the physical ranges.

    python tools/lab/random_gap_probe.py [--tiles 256] [--layouts 8]
"""
import argparse
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from proteus_amd import _capi            # noqa: E402
from proteus_amd.synth import SEED       # noqa: E402

T = 3660


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tiles', type=int, default=256)
    ap.add_argument('--layouts', type=int, default=8)
    ap.add_argument('--reps', type=int, default=4)
    ap.add_argument('--slack-gb', type=float, default=100.0)
    ap.add_argument('--quantum', type=int, default=2 << 20)
    ap.add_argument('--sweep', default='distance')
    ap.add_argument('--total-gb', type=float, default=0.0, help='arena size (default: batch + slack)')
    ap.add_argument('--in-base-gib', type=float, default=0.0, help='outpos sweep: where the packed input planes start')
    ap.add_argument('--out-from-gib', type=float, default=-1.0, help='outpos sweep: first start of the packed output planes (default: end of inputs)')
    ap.add_argument('--out-to-gib', type=float, default=-1.0)
    ap.add_argument('--out-step-gib', type=float, default=1.5)
    ap.add_argument('--span-gib', type=float, default=0.0, help='interleave sweep: minimum span of the write streams')
    ap.add_argument('--points-gib', default='', help='outpos sweep: explicit output starts, comma separated (after the range)')
    a = ap.parse_args()
    ctx = _capi.Context(0)
    params = _capi.default_params()
    stride = -(-T * T // 256) * 256
    geom = _capi.BatchGeom(a.tiles, T, T, stride)
    S = a.tiles * stride
    px = a.tiles * T * T
    sizes = [2 * S] * 6 + [S] + [2 * S] + [S] * 6            # 6 bands, fmask | diag, 6 u8 layers
    slack = int(a.slack_gb * 1e9)
    if a.sweep.startswith('separate'):
        # round 3: what a fresh process gets with TWO allocations (inputs, outputs) made in a given order
        order = a.sweep.split(':')[1] if ':' in a.sweep else 'in,out'
        bufs = {}
        for which in order.split(','):
            bufs[which] = ctx.malloc((sum(sizes[:7]) if which == 'in' else sum(sizes[7:])) + (1 << 20))
        pin, pout = _capi.PlanesIn(), _capi.PlanesOut()
        off = 0
        for i in range(6):
            pin.band[i] = bufs['in'].ptr + off
            off += sizes[i]
        pin.fmask = bufs['in'].ptr + off
        pout.diag = bufs['out'].ptr
        off = sizes[7]
        for name in ('wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
            setattr(pout, name, bufs['out'].ptr + off)
            off += S
        cb = ctx.malloc(a.tiles * 24)
        ctx.synth_batch(SEED, 0, geom, pin)
        for _ in range(2):
            ctx.classify_batch(params, geom, pin, pout, cb.ptr)
        ctx.synchronize()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(a.reps):
            ctx.classify_batch(params, geom, pin, pout, cb.ptr)
        ctx.record(e1)
        ctx.synchronize()
        ms = ctx.elapsed_ms(e0, e1) / a.reps
        print(json.dumps({'tiles': a.tiles, 'kind': 'separate allocations', 'order': order,
                          'in_ptr': hex(bufs['in'].ptr), 'out_ptr': hex(bufs['out'].ptr),
                          'GBps': round(px * 21 / ms / 1e6, 1)}))
        return
    arena = ctx.malloc(int(a.total_gb * 1e9) if a.total_gb else sum(sizes) + slack + (1 << 20))
    slack = arena.nbytes - sum(sizes) - (1 << 20)
    rng = random.Random(11)

    def bind(offsets):
        pin, pout = _capi.PlanesIn(), _capi.PlanesOut()
        for i in range(6):
            pin.band[i] = arena.ptr + offsets[i]
        pin.fmask = arena.ptr + offsets[6]
        pout.diag = arena.ptr + offsets[7]
        for k, name in enumerate(('wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud')):
            setattr(pout, name, arena.ptr + offsets[8 + k])
        return pin, pout

    def rate(pin, pout, counters):
        ctx.synth_batch(SEED, 0, geom, pin)
        for _ in range(2):
            ctx.classify_batch(params, geom, pin, pout, counters)
        ctx.synchronize()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(a.reps):
            ctx.classify_batch(params, geom, pin, pout, counters)
        ctx.record(e1)
        ctx.synchronize()
        ms = ctx.elapsed_ms(e0, e1) / a.reps
        ctx.destroy_event(e0)
        ctx.destroy_event(e1)
        return round(px * 21 / ms / 1e6, 1)

    counters_buf = ctx.malloc(a.tiles * 24)
    out = {'tiles': a.tiles, 'quantum': a.quantum, 'arena_bytes': arena.nbytes, 'arena_ptr': hex(arena.ptr), 'layouts': []}
    if a.sweep == 'outpos':
        # round 3: WHERE in the (first, large) allocation of a fresh process do the seven write streams run fast?
        # Inputs packed at --in-base-gib and generated once; the packed output region is moved through the arena.
        GiB = 1 << 30
        in_sz, out_sz = sum(sizes[:7]), sum(sizes[7:])
        in_base = int(a.in_base_gib * GiB)
        offs_in = [in_base + sum(sizes[:k]) for k in range(7)]
        pin0, _ = bind(offs_in + [0] * 7)
        ctx.synth_batch(SEED, 0, geom, pin0)
        starts = []
        lo = a.out_from_gib if a.out_from_gib >= 0 else (in_base + in_sz) / GiB
        hi = a.out_to_gib if a.out_to_gib >= 0 else (arena.nbytes - out_sz) / GiB
        x = lo
        while x <= hi + 1e-9:
            starts.append(x)
            x += a.out_step_gib
        starts += [float(v) for v in a.points_gib.split(',') if v]
        out.update({'inputs_at_GiB': [round(in_base / GiB, 2), round((in_base + in_sz) / GiB, 2)], 'outputs_GiB': round(out_sz / GiB, 2)})
        for st in starts:
            o0 = (int(st * GiB) + 255) & ~255
            if o0 + out_sz > arena.nbytes or (o0 < in_base + in_sz and o0 + out_sz > in_base):
                continue
            offs = offs_in + [o0 + sum(sizes[7:7 + j]) for j in range(7)]
            pin, pout = bind(offs)
            for _ in range(2):
                ctx.classify_batch(params, geom, pin, pout, counters_buf.ptr)
            ctx.synchronize()
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0)
            for _ in range(a.reps):
                ctx.classify_batch(params, geom, pin, pout, counters_buf.ptr)
            ctx.record(e1)
            ctx.synchronize()
            ms = ctx.elapsed_ms(e0, e1) / a.reps
            ctx.destroy_event(e0)
            ctx.destroy_event(e1)
            out['layouts'].append({'out_start_GiB': round(st, 2), 'GBps': round(px * 21 / ms / 1e6, 1)})
        print(json.dumps(out))
        return

    def packed(order):
        off, offsets = 0, [0] * 14
        for k in order:
            offsets[k] = off
            off += sizes[k]
        return offsets

    # packed, inputs then outputs (what DeviceBatch does)
    out['layouts'].append({'kind': 'packed', 'GBps': rate(*bind(packed(range(14))), counters_buf.ptr)})
    # packed, planes in a random order
    for _ in range(2):
        order = list(range(14))
        rng.shuffle(order)
        out['layouts'].append({'kind': 'packed, shuffled order', 'order': order,
                               'GBps': rate(*bind(packed(order)), counters_buf.ptr)})
    # inputs packed, outputs with random gaps (the slack split at random among the 7 output planes)
    for _ in range(a.layouts):
        cuts = sorted(rng.randrange(0, slack // a.quantum) for _ in range(7))
        gaps = [cuts[0]] + [cuts[i] - cuts[i - 1] for i in range(1, 7)]
        off, offsets = 0, [0] * 14
        for k in range(7):
            offsets[k] = off
            off += sizes[k]
        for j, k in enumerate(range(7, 14)):
            off += gaps[j] * a.quantum
            offsets[k] = off
            off += sizes[k]
        out['layouts'].append({'kind': 'outputs at random gaps', 'gaps_MB': [g * a.quantum >> 20 for g in gaps],
                               'GBps': rate(*bind(offsets), counters_buf.ptr)})
    out['layouts'].append({'kind': 'packed (again)', 'GBps': rate(*bind(packed(range(14))), counters_buf.ptr)})

    def gapped(gaps_bytes, in_gaps_bytes=None):
        off, offsets = 0, [0] * 14
        for k in range(7):
            if in_gaps_bytes:
                off += in_gaps_bytes[k]
            offsets[k] = off
            off += sizes[k]
        for j, k in enumerate(range(7, 14)):
            off += gaps_bytes[j]
            offsets[k] = off
            off += sizes[k]
        assert off <= arena.nbytes, (off, arena.nbytes)
        return offsets

    GB = 1 << 30
    if a.sweep == 'interleave':
        # round 3: the write streams spread over the arena by ALTERNATING output and input planes (no memory
        # wasted) vs packed inputs-then-outputs, both moved through the arena; --span-gib stretches the
        # interleaved layout with equal gaps in front of the output planes until the write streams span that much
        il = [7, 0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6]

        def interleaved(span_gib):
            natural = sum(sizes[k] for k in il[:-2])            # start of the first to start of the last output plane
            extra = max(0, int(span_gib * GB) - natural) // 6 & ~255
            off, offsets = 0, [0] * 14
            for n, k in enumerate(il):
                if k >= 7 and n:
                    off += extra
                offsets[k] = off
                off += sizes[k]
            return offsets, off
        bases = [float(v) for v in a.points_gib.split(',') if v] or [0.0]
        for b_gib in bases:
            row = {'base_GiB': b_gib}
            for name, (offs, end) in (('packed', (packed(range(14)), sum(sizes))),
                                      ('interleaved', interleaved(0)),
                                      ('interleaved_span', interleaved(a.span_gib))):
                if name == 'interleaved_span' and not a.span_gib:
                    continue
                b = int(b_gib * GB)
                if b + end > arena.nbytes:
                    continue
                row[name] = rate(*bind([o + b for o in offs]), counters_buf.ptr)
                row[name + '_bytes'] = end
            out['layouts'].append(row)
        print(json.dumps(out))
        return
    if a.sweep == 'period':
        # the output region moved away from the input region in 3 GB steps: is the rate periodic in the distance?
        for g_gb in range(0, 130, 3):
            out['layouts'].append({'kind': 'one gap between inputs and outputs', 'gap_GB': g_gb,
                                   'outputs_start_GB': round((sum(sizes[:7]) + g_gb * GB) / GB, 1),
                                   'GBps': rate(*bind(gapped([g_gb * GB] + [0] * 6)), counters_buf.ptr)})
        # the whole packed layout moved through the arena: does the absolute position matter?
        for b_gb in (0, 9, 18, 27, 36, 54, 72, 100):
            offs = [o + b_gb * GB for o in packed(range(14))]
            out['layouts'].append({'kind': 'packed layout moved as a whole', 'base_GB': b_gb, 'GBps': rate(*bind(offs), counters_buf.ptr)})
    if a.sweep == 'distance':
        # one gap between the input region and the output region
        for g_gb in (12, 16, 20, 24, 28, 32, 40, 48, 64, 96):
            out['layouts'].append({'kind': f'one gap of {g_gb} GB between inputs and outputs',
                                   'GBps': rate(*bind(gapped([g_gb * GB + (6 << 20)] + [0] * 6)), counters_buf.ptr)})
        # every plane the same distance from the next
        for g_gb in (1, 2, 4, 6, 8, 10, 12):
            out['layouts'].append({'kind': f'all 14 planes at uniform gaps of {g_gb} GB',
                                   'GBps': rate(*bind(gapped([g_gb * GB + (6 << 20)] * 7, [0] + [g_gb * GB + (6 << 20)] * 6)), counters_buf.ptr)})
        # outputs only, uniform
        for g_gb in (4, 5, 6, 8, 12):
            out['layouts'].append({'kind': f'outputs at uniform gaps of {g_gb} GB',
                                   'GBps': rate(*bind(gapped([g_gb * GB + (6 << 20)] * 7)), counters_buf.ptr)})
    out['layouts'].append({'kind': 'packed (last)', 'GBps': rate(*bind(packed(range(14))), counters_buf.ptr)})
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
