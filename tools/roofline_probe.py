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
    ap.add_argument('--quick', action='store_true', help='only the main shapes')
    ap.add_argument('--store-policies', action='store_true', help='A/B of store cache policies (global instr)')
    ap.add_argument('--policies', action='store_true', help='cache-policy grid of the fused shape')
    ap.add_argument('--tile-align', type=int, default=256, help='1 = contiguous tiles')
    a = ap.parse_args()
    ctx = _capi.Context(0)
    batch = _capi.DeviceBatch(ctx, a.tiles, a.size, a.size, masks=a.masks, tile_align=a.tile_align)
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
                                                     variant, tile_stride=batch.tile_stride), a.reps)
            out[f'flat 2-stream copy nt={int(bool(variant & 2))}'] = round(px * 21 / (sum(ms) / len(ms)) / 1e6, 1)
        for variant in (256 | 1024, 256 | 1024 | 2):
            ms = timed(ctx, lambda: ctx.stream_probe(a.tiles, batch.n_pixels, batch.pin, batch.pout,
                                                     variant, tile_stride=batch.tile_stride), a.reps)
            npx = (px // 4096) * 4096
            out[f'steady 13:8 two-stream copy nt={int(bool(variant & 2))}'] = round(npx * 21 / (sum(ms) / len(ms)) / 1e6, 1)

        for lg in (0, 1, 2, 3):
            for nt in (0, 2):
                variant = 256 | 1024 | 4 | (lg << 4) | nt
                ms = timed(ctx, lambda: ctx.stream_probe(a.tiles, batch.n_pixels, batch.pin, batch.pout,
                                                         variant, tile_stride=batch.tile_stride), a.reps)
                nb = (px * 13 // 16 // 832) >> lg
                out[f'steady 13:8 copy, {1 << lg} word(s) per thread nt={nt >> 1}'] = \
                    round((nb << lg) * 832 * 16 * 21 / 13 / (sum(ms) / len(ms)) / 1e6, 1)

        def run(label, variant, nbytes):
            ms = timed(ctx, lambda: ctx.stream_probe(a.tiles, batch.n_pixels, batch.pin, batch.pout,
                                                     variant, tile_stride=batch.tile_stride), a.reps)
            out[label] = round(px * nbytes / (sum(ms) / len(ms)) / 1e6, 1)
        if a.store_policies:
            # interleaved rounds: same-process A/B of the store policies with global instructions
            names = ('asm nt', 'asm sc1 nt', 'asm sc0 sc1 nt', 'asm sc1', 'builtin nt')
            acc = {n: [] for n in names}
            for _ in range(7):
                for sp, n in enumerate(names):
                    ms = timed(ctx, lambda: ctx.stream_probe(a.tiles, batch.n_pixels, batch.pin, batch.pout,
                                                             (1 << 24) | (sp << 2), tile_stride=batch.tile_stride), 4)
                    acc[n].append(px * 21 / (sum(ms) / len(ms)) / 1e6)
            for n in names:
                v = sorted(acc[n])
                out[f'store policy [{n}]'] = {'median': round(v[len(v) // 2], 1), 'min': round(v[0], 1), 'max': round(v[-1], 1)}
            print(json.dumps(out, indent=1))
            return
        if a.policies:
            names = ('plain', 'nt', 'sc1', 'sc0 sc1', 'sc1 nt', 'sc0 sc1 nt', 'sc0', 'sc0 nt')
            for li, ln in enumerate(names):
                for si, sn in enumerate(names):
                    run(f'policy probe: loads [{ln}] stores [{sn}]', (1 << 23) | (li << 2) | (si << 5), 21)
            print(json.dumps(out, indent=1))
            return
        for sel, ch in enumerate((2048, 8192, 32768, 131072)):
            for nt in (0, 2):
                run(f'record layout CHPX={ch} nt={nt >> 1}', (1 << 22) | (sel << 2) | nt, 21)
        run('fused shape ppt=8 nt=1 (as is)', 2, 21)
        run('thin-4 shape (4 px/thread, 3.25 loads + 2 stores per wave) nt=0', 524288 | 8, 21)
        run('thin-4 shape (4 px/thread, 3.25 loads + 2 stores per wave) nt=1', 524288 | 8 | 2, 21)
        run('warp-specialised (4 fat waves) nt=1', 524288 | 2, 21)
        run('warp-specialised THIN (16 waves, <= 2 loads + 1 store each) nt=0', 524288 | 4, 21)
        run('warp-specialised THIN (16 waves, <= 2 loads + 1 store each) nt=1', 524288 | 4 | 2, 21)
        run('fused shape ppt=8 nt=1 xcdmap', 2 | 2048, 21)
        run('fused shape ppt=8 nt=1 block512', 2 | 4096, 21)
        if a.quick:
            print(json.dumps(out, indent=1))
            return
        for nt in (0, 2):
            run(f'14 planes, one plane per block nt={nt >> 1}', 32768 | nt, 21)
            run(f'one plane per wave (7-wave blocks over 4096 px) nt={nt >> 1}', 262144 | nt, 21)
            run(f'warp-specialised LDS-DMA in, plane-run stores out nt={nt >> 1}', 524288 | nt, 21)
            run(f'role split by block (7-plane readers / 7-plane writers) nt={nt >> 1}', 65536 | nt, 21)
            run(f'role split by wave inside block nt={nt >> 1}', 65536 | 4 | nt, 21)
            run(f'role split by block, writers use plane-run stores nt={nt >> 1}', 65536 | 8 | nt, 21)
        # chunk-interleaved layout (needs total px multiple of the chunk: use a prefix)
        for sel, ch in ((0, 4096), (1, 65536), (2, 1048576), (3, 16777216)):
            for nt in (0, 2):
                npx = (px // ch) * ch
                ms = timed(ctx, lambda: ctx.stream_probe(1, npx, batch.pin, batch.pout,
                                                         131072 | (sel << 2) | nt), a.reps)
                out[f'chunk-interleaved planes CH={ch} nt={nt >> 1}'] = round(npx * 21 / (sum(ms) / len(ms)) / 1e6, 1)
        for lb in (0, 1, 2):
            for nt in (0, 2):
                run(f'staged probe block={256 << lb} nt={nt >> 1}', 16384 | (lb << 2) | nt, 21)
        run('probe ppt=8 nt=1 (direct stores)', 2, 21)
        for ppt16 in (0, 1):
            for nt in (0, 2):
                tag = f'ppt={16 if ppt16 else 8} nt={nt >> 1}'
                run(f'fused shape {tag} (as is)', ppt16 | nt, 21)
                run(f'fused shape {tag}, all stores lane-contiguous 16 B', ppt16 | nt | (1 << 21), 21)
                run(f'fused shape {tag} write-only (as is)', ppt16 | nt | (2 << 9), 8)
                run(f'fused shape {tag} write-only, all stores lane-contiguous 16 B', ppt16 | nt | (1 << 9) | (1 << 21), 8)
        for pi, P in enumerate((1, 2, 3, 6)):
            for ri, R in enumerate((1, 2, 4, 8)):
                for nt in (0, 2):
                    run(f'write grid: {P} planes per wave x {R} KiB runs nt={nt >> 1}',
                        1048576 | (pi << 2) | (ri << 4) | nt, 6)
        for wm, name in ((0, 'flat 1 stream'), (1, '7 planes, plane per block'), (2, '7 planes, plane per wave')):
            for nt in (0, 2):
                run(f'write-only {name} nt={nt >> 1}', 8192 | (wm << 2) | nt, 8)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
