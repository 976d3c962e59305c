#!/usr/bin/env python3
"""Soak test: random parameters x random rasters (synthetic tiles, uniform int16 noise, values
hugging the thresholds) through dswx_classify_host, every layer and the counters compared with
the scalar C oracle.  Exit code 1 and a JSON description of the first mismatch on failure.

    python tests/helpers/fuzz_parity.py [--iters 300] [--seed 1] [--variant N] [--device-batch] [--pinned] [--odd-planes]

(Lives under tests/ because it uses the oracle, which only tests/, smoke() and bench.py's CPU baseline may do;
its name keeps pytest from collecting it.)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle                         # noqa: E402  (checker)
from proteus_amd import _capi                       # noqa: E402
from proteus_amd.synth import synth_tile            # noqa: E402
from tests.test_gpu_parity import _random_case, ALL_LAYERS   # noqa: E402


def device_batch_soak(ctx, rng, a, kernels):
    """Device-resident batches with random geometry: tile count, ragged tile sizes, tile-stride
    alignment (1 / 8 / 16 / 64 / 256 px), one allocation / one per output plane / a VMM-backed sliding range (sometimes
    placed first), optional planes and layers,
    counters on / off, 'mask' / 'ignore' / 'cover'; every tile against the oracle."""
    from oracle import dswx_oracle as o
    from tests.test_c_oracle import NAME
    for it in range(a.iters):
        cs = _random_case(rng)
        n_tiles = int(rng.integers(1, 6))
        h, w = int(rng.integers(1, 120)), int(rng.integers(1, 150))
        align = int(rng.choice([1, 8, 16, 64, 256]))
        masks = bool(rng.integers(2))
        form = str(rng.choice(['arena', 'arena', 'separate', 'sliding']))
        extra = tuple(x for x in ('wtr1_aerosol', 'browse') if rng.integers(2))
        batch = _capi.DeviceBatch(ctx, n_tiles, h, w, masks=masks, extra_layers=extra, tile_align=align,
                                  separate_outputs=form == 'separate', sliding_outputs=form == 'sliding')
        batch.synth(777 + it, tile0=it)
        mode = str(rng.choice(['mask', 'ignore', 'cover']))
        p = _capi.make_params(
            cs['thr'], clip_negative_reflectance=cs['clip'], mask_adjacent_to_cloud_mode=mode,
            apply_aerosol_class_remapping=cs['aerosol'], aerosol_fmask_values=cs['lists'],
            collapse_wtr_classes=cs['collapse'],
            aerosol_max_nir=None if mode == 'cover' else cs['aer_nir'])   # the numpy oracle fixes it at 1000
        use_counters = bool(rng.integers(4))
        if form == 'sliding' and rng.integers(2):       # place it first: the planes move inside a wider range
            batch.place_slide(p, slack_bytes=int(rng.integers(1, 9)) << 20, step_bytes=1 << 20,
                              spread_gaps=int(rng.integers(0, 3)), refine_passes=int(rng.integers(0, 2)), launches=1)
        elif form == 'separate' and rng.integers(2):
            batch.place_search(p, candidates=2, launches=1)
        batch.classify(p, counters=use_counters)
        ctx.synchronize()
        key = ctx.last_kernel_info().split(' grid')[0]
        kernels[key] = kernels.get(key, 0) + 1
        cnt = batch.read_counters() if use_counters else None
        for t in range(n_tiles):
            bands = [batch.read_tile(b, t) for b in _capi.BAND_NAMES]
            fm = batch.read_tile('fmask', t)
            kw = {m: batch.read_tile(m, t) for m in ('land', 'shad', 'ocean')} if masks else {}
            if mode == 'cover':
                with np.errstate(all='ignore'):
                    e = o.classify_tile(bands, fm, o.Thresholds(**cs['thr']), landcover=kw.get('land'),
                                        shadow=kw.get('shad'), ocean_mask=kw.get('ocean'),
                                        mask_adjacent_to_cloud_mode='cover', apply_aerosol=cs['aerosol'],
                                        aerosol_fmask_values=cs['lists'], clip_negative_reflectance=cs['clip'],
                                        collapse=cs['collapse'])
                exp = {k: e[layer] for layer, k in NAME.items()}
                ec = e['counters']
                exp_cnt = [ec['n_valid'], ec['n_cloud_and_valid'], ec['n_not_ocean']]
            else:
                exp = c_oracle.classify(p, bands, fm, **kw)
                exp_cnt = exp['counters'].tolist()
            layers = ['diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'] + [x for x in extra if x == 'wtr1_aerosol']
            bad = [k for k in layers if not np.array_equal(batch.read_tile(k, t), exp[k])]
            if use_counters and cnt[t].tolist() != exp_cnt:
                bad.append('counters')
            if bad:
                print(json.dumps({'ok': False, 'iteration': it, 'tile': t, 'geom': [n_tiles, h, w, align],
                                  'masks': masks, 'mode': mode, 'extra': extra, 'layers': bad,
                                  'form': form, 'kernel': ctx.last_kernel_info()},
                                 default=str))
                return 1
        batch.free()
    print(json.dumps({'ok': True, 'iterations': a.iters, 'seed': a.seed, 'kernels': kernels}))
    return 0


def odd_planes_soak(ctx, rng, a, kernels):
    """Round 6: planes carved out of one arena at RANDOM addresses (int16 planes at any even address, byte planes anywhere),
    random tile stride >= H W, several tiles, every mode incl. 'cover', masks, optional layers, both chains: the direct
    kernel's unaligned accesses (and 'cover' stage 3's), the generic kernel's tails -- every tile against the oracles,
    and no byte outside the planes touched."""
    from oracle import dswx_oracle as o
    from tests.test_c_oracle import NAME
    for it in range(a.iters):
        cs = _random_case(rng)
        n = int(rng.integers(1, 5))
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 130))
        P = h * w
        stride = P + int(rng.choice([0, 0, 1, 3, 5, 8, 13]))
        masks = bool(rng.integers(2))
        mode = str(rng.choice(['mask', 'ignore', 'cover']))
        scaled = None
        if mode != 'cover' and rng.integers(5) == 0:
            scaled = [(float(rng.choice([1.0, 0.5, 1e-4])), float(rng.choice([0.0, 0.5, -100.0]))) for _ in range(6)]
        p = _capi.make_params(cs['thr'], clip_negative_reflectance=cs['clip'], mask_adjacent_to_cloud_mode=mode,
                              apply_aerosol_class_remapping=cs['aerosol'], aerosol_fmask_values=cs['lists'],
                              collapse_wtr_classes=cs['collapse'], aerosol_max_nir=None if mode == 'cover' else cs['aer_nir'],
                              offset_and_scale=scaled)
        tiles = [synth_tile(9000 + 7 * it + t, h, w, with_masks=True) for t in range(n)]
        in_names = list(_capi.BAND_NAMES) + ['fmask'] + (['land', 'shad', 'ocean'] if masks else [])
        out_names = ['diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'] + (['wtr1_aerosol'] if rng.integers(2) else [])
        sizes = {k: n * stride * (2 if (k in _capi.BAND_NAMES or k == 'diag') else 1) for k in in_names + out_names}
        off, where = int(rng.integers(0, 16)), {}
        for k in in_names + out_names:
            two = k in _capi.BAND_NAMES or k == 'diag'
            off += int(rng.integers(0, 9))
            if two:
                off += off % 2
            where[k] = off
            off += sizes[k]
        off += (-off) % 8
        cnt_off = off
        total = off + n * 24 + 64
        host = np.full(total, 0x5A, np.uint8)
        for k in in_names:
            for t in range(n):
                src = tiles[t]['bands'][_capi.BAND_NAMES.index(k)] if k in _capi.BAND_NAMES else tiles[t][k]
                raw = np.ascontiguousarray(src).reshape(-1).view(np.uint8)
                esz = 2 if k in _capi.BAND_NAMES else 1
                host[where[k] + t * stride * esz: where[k] + t * stride * esz + raw.size] = raw
        arena = ctx.malloc(total)
        arena.upload(host)
        pin, pout = _capi.PlanesIn(), _capi.PlanesOut()
        for i, k in enumerate(_capi.BAND_NAMES):
            pin.band[i] = arena.ptr + where[k]
        for k in in_names[6:]:
            setattr(pin, k, arena.ptr + where[k])
        for k in out_names:
            setattr(pout, k, arena.ptr + where[k])
        geom = _capi.BatchGeom(n, h, w, stride)
        ctx.classify_batch(p, geom, pin, pout, arena.ptr + cnt_off)
        ctx.synchronize()
        key = ctx.last_kernel_info().split(' grid')[0]
        kernels[key] = kernels.get(key, 0) + 1
        got = arena.download(np.uint8, total)
        arena.free()
        bad = []
        touched = np.ones(total, bool)
        for k in in_names:
            touched[where[k]: where[k] + sizes[k]] = False
        for k in out_names:
            esz = 2 if k == 'diag' else 1
            for t in range(n):
                touched[where[k] + t * stride * esz: where[k] + (t * stride + P) * esz] = False
        touched[cnt_off: cnt_off + n * 24] = False
        if (got[touched] != host[touched]).any() or not np.array_equal(got[~touched][:0], host[~touched][:0]):
            bad.append('bytes outside the planes')
        for k in in_names:
            if not np.array_equal(got[where[k]: where[k] + sizes[k]], host[where[k]: where[k] + sizes[k]]):
                bad.append('input plane ' + k + ' modified')
        cnt = got[cnt_off: cnt_off + n * 24].view(np.int64).reshape(n, 3)
        for t in range(n):
            st = tiles[t]
            kw = {m: st[m] for m in ('land', 'shad', 'ocean')} if masks else {}
            if mode == 'cover':
                with np.errstate(all='ignore'):
                    e = o.classify_tile(st['bands'], st['fmask'], o.Thresholds(**cs['thr']), landcover=kw.get('land'),
                                        shadow=kw.get('shad'), ocean_mask=kw.get('ocean'), mask_adjacent_to_cloud_mode='cover',
                                        apply_aerosol=cs['aerosol'], aerosol_fmask_values=cs['lists'],
                                        clip_negative_reflectance=cs['clip'], collapse=cs['collapse'])
                exp = {kk: e[layer] for layer, kk in NAME.items()}
                ec = e['counters']
                exp_cnt = [ec['n_valid'], ec['n_cloud_and_valid'], ec['n_not_ocean']]
            else:
                exp = c_oracle.classify(p, st['bands'], st['fmask'], **kw)
                exp_cnt = exp['counters'].tolist()
            for k in out_names:
                esz = 2 if k == 'diag' else 1
                g = got[where[k] + t * stride * esz: where[k] + (t * stride + P) * esz]
                g = g.view(np.uint16) if k == 'diag' else g
                if not np.array_equal(g.reshape(h, w), exp[k]):
                    bad.append(f'{k}[{t}]')
            if cnt[t].tolist() != exp_cnt:
                bad.append(f'counters[{t}]')
        if bad:
            print(json.dumps({'ok': False, 'iteration': it, 'geom': [n, h, w, stride], 'masks': masks, 'mode': mode,
                              'scaled': scaled, 'what': bad[:8], 'kernel': ctx.last_kernel_info()}, default=str))
            return 1
    print(json.dumps({'ok': True, 'iterations': a.iters, 'seed': a.seed, 'kernels': kernels}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=300)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--variant', default=None, help='force a product kernel for this run (libdswx_lab.so switch: 0 direct, 3 table-driven)')
    ap.add_argument('--device-batch', action='store_true', help='soak the device-resident batch entry instead')
    ap.add_argument('--pinned', action='store_true', help='inputs in page-locked arrays: the zero-copy host path')
    ap.add_argument('--odd-planes', action='store_true', help='planes at random (odd) addresses and strides through dswx_classify_batch')
    a = ap.parse_args()
    ctx = _capi.Context(0)
    if a.variant is not None:
        ctx.lab_configure(fused_variant=int(a.variant))
    rng = np.random.default_rng(a.seed)
    kernels = {}
    if a.device_batch:
        return device_batch_soak(ctx, rng, a, kernels)
    if a.odd_planes:
        return odd_planes_soak(ctx, rng, a, kernels)
    for it in range(a.iters):
        cs = _random_case(rng)
        kind = it % 4
        h, w = int(rng.integers(1, 200)), int(rng.integers(1, 260))
        if kind == 3:
            h, w = int(rng.integers(200, 700)), 8 * int(rng.integers(30, 120))
        s = synth_tile(5000 + it, h, w, with_masks=True)
        bands = [b.copy() for b in s['bands']]
        fmask = s['fmask'].copy()
        if kind == 1:          # uniform int16 noise incl. the extremes, random Fmask bytes
            bands = [rng.integers(-32768, 32768, size=(h, w)).astype(np.int16) for _ in range(6)]
            fmask = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
        elif kind == 2:        # values hugging the integer thresholds and the quotient ties
            t = cs['thr']
            for b, key in zip(bands, ('pswt_2_blue', None, None, 'pswt_1_nir', 'pswt_1_swir1', 'pswt_2_swir2')):
                if key is not None and abs(t[key]) < 30000:
                    b[...] = (int(t[key]) + rng.integers(-2, 3, size=(h, w))).astype(np.int16)
            g = rng.integers(1, 3000, size=(h, w))
            bands[1][...] = g
            bands[4][...] = np.clip((g * (1 - t['wigt']) / (1 + t['wigt'] + 1e-9)).astype(np.int64)
                                    + rng.integers(-1, 2, size=(h, w)), -32768, 32767)
        # one case in six: flag_offset_and_scale_inputs, the float32 chain (scales of 1 keep the random thresholds in play;
        # the uniform-noise and threshold-hugging rasters then probe float32 rounding at the thresholds)
        scaled = None
        if rng.integers(6) == 0:
            scaled = [(float(rng.choice([1.0, 1.0, 0.5, 1e-4, 3.0])), float(rng.choice([0.0, 0.0, 0.5, -100.0, 7.25])))
                      for _ in range(6)]
        p = _capi.make_params(
            cs['thr'], band_fills=cs['fills'], fmask_fill=cs['fmask_fill'],
            clip_negative_reflectance=cs['clip'], mask_adjacent_to_cloud_mode=cs['mode'],
            apply_aerosol_class_remapping=cs['aerosol'], aerosol_fmask_values=cs['lists'],
            collapse_wtr_classes=cs['collapse'], aerosol_max_nir=cs['aer_nir'], offset_and_scale=scaled)
        kw = {k: s[k] for k in ('land', 'shad', 'ocean') if cs[k]}
        if a.pinned:
            def pin(x):
                q = ctx.pinned_empty(x.shape, x.dtype)
                q[...] = x
                return q
            bands, fmask, kw = [pin(b) for b in bands], pin(fmask), {k: pin(v) for k, v in kw.items()}
        got = ctx.classify_host(bands, fmask, p, **kw)
        if a.pinned and 'zero copy' not in ctx.last_kernel_info():
            print(json.dumps({'ok': False, 'iteration': it, 'why': 'not the zero-copy path', 'kernel': ctx.last_kernel_info()}))
            return 1
        kernels[ctx.last_kernel_info().split(' ')[0]] = kernels.get(ctx.last_kernel_info().split(' ')[0], 0) + 1
        exp = c_oracle.classify(p, bands, fmask, **kw)
        bad = [k for k in ALL_LAYERS if not np.array_equal(got[k], exp[k])]
        if got['counters'][0].tolist() != exp['counters'].tolist():
            bad.append('counters')
        if bad:
            print(json.dumps({'ok': False, 'iteration': it, 'kind': kind, 'shape': [h, w], 'layers': bad, 'scaled': scaled,
                              'case': {k: (v if not isinstance(v, dict) else v) for k, v in cs.items()}},
                             default=str))
            return 1
    print(json.dumps({'ok': True, 'iterations': a.iters, 'seed': a.seed, 'kernels': kernels}))
    return 0


if __name__ == '__main__':
    sys.exit(main())
