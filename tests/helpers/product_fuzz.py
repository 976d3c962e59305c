#!/usr/bin/env python3
"""Soak of the WHOLE product path with random option sets: random ragged tile sizes, sensors, masks (none / LAND / SHAD /
ocean, as arrays), cloud modes, the float32 chain, and a random subset of every output the reference can write -- the seven
layers, LAND / SHAD layers, the two Float32 composites, the browse image (random size), the ten-band Byte file -- through
proteus_amd.dswx_hls.generate_dswx_layers, K calls side by side on threads of this process (one HIP context, the engine's
pools).  Every file that comes out is checked: layers against the numpy oracle, first overview level against the row-by-row
NEAREST restatement (oracle/cog_oracle.py), composites against the reference's statement, the multi-band file band by
band, every file against the COG layout rules.  Prints one JSON object; exit code 1 on the first mismatch.

    python tests/helpers/product_fuzz.py [--cases 120] [--threads 4] [--seed 1]

(Lives under tests/ because it uses the oracle; its name keeps pytest from collecting it.)"""
import argparse
import json
import logging
import os
import struct
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_synthetic_hls as synth_hls                  # noqa: E402
from oracle import cog_oracle, dswx_oracle as o         # noqa: E402  (checkers)
from proteus_amd import dswx_hls as D, geotiff          # noqa: E402

LAYER_ARGS = (('WTR', 'output_interpreted_band'), ('BWTR', 'output_binary_water'), ('CONF', 'output_confidence_layer'),
              ('DIAG', 'output_diagnostic_layer'), ('WTR-1', 'output_non_masked_dswx'), ('WTR-2', 'output_shadow_masked_dswx'),
              ('CLOUD', 'output_cloud_layer'))


class Mismatch(Exception):
    pass


def one_case(case, root):
    rng = np.random.default_rng(1000 + case)
    d = os.path.join(root, f'c{case}')
    h = int(rng.integers(1, 70)) * 8 + int(rng.integers(0, 8)) if rng.random() < 0.8 else int(rng.integers(1, 9))
    size = max(h, 2)
    # two cases in five: a nodata wedge at the top of the tile (a granule at a swath edge; up to the whole tile); band files in small tiles
    fill_rows = float(rng.choice([0.0, 0.0, 0.3, 0.7, 1.0])) if size >= 16 else 0.0
    file_tile = int(rng.choice([16, 64, 128, 512]))
    _, files, _, s = synth_hls.make(d, sensor=('L30', 'S30')[case % 2], size=size, tile=5000 + case, masks=True,
                                    fill_rows=fill_rows, file_tile=file_tile)
    kw, okw, desc = {}, {}, {'case': case, 'size': size, 'fill_rows': fill_rows, 'file_tile': file_tile}
    if rng.random() < 0.5:
        kw['landcover_mask'], okw['landcover'] = s['land'], s['land']
    if rng.random() < 0.5:
        kw['shadow_layer'], okw['shadow'] = s['shad'].astype(bool), s['shad']
    if rng.random() < 0.4:
        kw['ocean_mask'], kw['apply_ocean_masking'], okw['ocean_mask'] = s['ocean'], True, s['ocean']
    mode = str(rng.choice(['mask', 'ignore', 'cover']))
    kw['mask_adjacent_to_cloud_mode'] = okw['mask_adjacent_to_cloud_mode'] = mode
    scaled = rng.random() < 0.25
    if scaled:
        kw['flag_offset_and_scale_inputs'] = True
        okw['offset_and_scale'] = [(0.0001, 0.0)] * 6
    outs = {}
    for layer, arg in LAYER_ARGS:
        if rng.random() < 0.6:
            outs[layer] = kw[arg] = os.path.join(d, f'{layer}.tif')
    if 'landcover_mask' in kw and rng.random() < 0.5:
        outs['LAND'] = kw['output_landcover'] = os.path.join(d, 'LAND.tif')
    if 'shadow_layer' in kw and rng.random() < 0.5:
        outs['SHAD'] = kw['output_shadow_layer'] = os.path.join(d, 'SHAD.tif')
    if rng.random() < 0.4:
        outs['rgb'] = kw['output_rgb_file'] = os.path.join(d, 'rgb.tif')
    if rng.random() < 0.3:
        outs['irgb'] = kw['output_infrared_rgb_file'] = os.path.join(d, 'irgb.tif')
    if rng.random() < 0.4:
        outs['browse'] = kw['output_browse_image'] = os.path.join(d, 'browse.png')
        kw['browse_image_height'], kw['browse_image_width'] = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    multiband = os.path.join(d, 'product.tif') if rng.random() < 0.4 else None
    if not outs and not multiband:
        outs['WTR'] = kw['output_interpreted_band'] = os.path.join(d, 'WTR.tif')
    desc.update(mode=mode, scaled=scaled, masks=sorted(okw.keys() & {'landcover', 'shadow', 'ocean_mask'}), outputs=sorted(outs), multiband=bool(multiband))
    if not D.generate_dswx_layers(files, multiband, scratch_dir=os.path.join(d, 'scratch'), **kw):
        raise Mismatch(f'{desc}: generate_dswx_layers returned False')
    exp = o.classify_tile(s['bands'], s['fmask'], **okw)
    exp['LAND'], exp['SHAD'] = s['land'], s['shad']
    checked = 0
    for layer, path in outs.items():
        if layer in ('rgb', 'irgb', 'browse'):
            continue
        arr, _ = geotiff.read_geotiff(path)
        if not np.array_equal(arr, exp[layer]):
            raise Mismatch(f'{desc}: layer {layer}: {int(np.count_nonzero(arr != exp[layer]))} pixels differ')
        if geotiff.validate_cog(path):
            raise Mismatch(f'{desc}: layer {layer}: {geotiff.validate_cog(path)}')
        if arr.shape != (1, 1):
            ovr, _ = geotiff.read_geotiff(path, overview=0)
            if not np.array_equal(ovr, cog_oracle.nearest_overview(np.asarray(exp[layer]), 4)):
                raise Mismatch(f'{desc}: layer {layer}: first overview level')
        checked += 1
    valid = exp['DIAG'] != 65535
    for key, idx in (('rgb', (2, 1, 0)), ('irgb', (4, 3, 2))):
        if key in outs:
            got, _ = geotiff.read_geotiff(outs[key])
            want = cog_oracle.rgb_planes([s['bands'][i] for i in idx], exp['DIAG'], [0.0001] * 3, [0.0] * 3)
            if got.shape != want.shape or not np.array_equal(got, want, equal_nan=True) or np.isnan(got[:, valid]).any() or \
                    not np.isnan(got[:, ~valid]).all():
                raise Mismatch(f'{desc}: composite {key}')
            if geotiff.validate_cog(outs[key]):
                raise Mismatch(f'{desc}: composite {key}: {geotiff.validate_cog(outs[key])}')
            geotiff.read_geotiff(outs[key], overview=0)            # the CUBICSPLINE level is there and decodes
            checked += 1
    if 'browse' in outs:
        raw = open(outs['browse'], 'rb').read()
        w, hh = struct.unpack('>II', raw[16:24])
        if raw[:8] != b'\x89PNG\r\n\x1a\n' or (hh, w) != (kw['browse_image_height'], kw['browse_image_width']):
            raise Mismatch(f'{desc}: browse PNG {hh} x {w}')
        if geotiff.validate_cog(outs['browse'].replace('.png', '.tif')):
            raise Mismatch(f'{desc}: browse GeoTIFF layout')
        checked += 1
    if multiband:
        stack, info = geotiff.read_geotiff(multiband)
        fill = np.full(exp['WTR'].shape, 255, np.uint8)
        want = [exp['WTR'], exp['BWTR'], np.minimum(exp['DIAG'], 255).astype(np.uint8), exp['WTR-1-AEROSOL'], exp['WTR-2'],
                s['land'] if 'landcover' in okw else fill, s['shad'] if 'shadow' in okw else fill, exp['CLOUD'], fill]
        for i, wnt in enumerate(want):
            if not np.array_equal(stack[i], wnt):
                raise Mismatch(f'{desc}: multi-band file, band {i + 1}')
        if stack[9].any() or info.bands != 10 or geotiff.validate_cog(multiband):
            raise Mismatch(f'{desc}: multi-band file layout')
        checked += 1
    return checked, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=120)
    ap.add_argument('--threads', type=int, default=4)
    ap.add_argument('--seed', type=int, default=1)
    a = ap.parse_args()
    logging.getLogger('dswx_hls').setLevel(logging.ERROR)
    D.get_context(0)
    t0 = time.perf_counter()
    seen = {'mode': {}, 'outputs': {}, 'scaled': 0, 'multiband': 0, 'with_nodata_rows': 0}
    with tempfile.TemporaryDirectory() as root:
        try:
            with ThreadPoolExecutor(a.threads) as ex:
                results = list(ex.map(lambda c: one_case(c, root), range(a.seed * 100000, a.seed * 100000 + a.cases)))
        except Mismatch as e:
            print(json.dumps({'ok': False, 'why': str(e)}))
            return 1
    for _, desc in results:
        seen['mode'][desc['mode']] = seen['mode'].get(desc['mode'], 0) + 1
        for k in desc['outputs']:
            seen['outputs'][k] = seen['outputs'].get(k, 0) + 1
        seen['scaled'] += desc['scaled']
        seen['multiband'] += desc['multiband']
        seen['with_nodata_rows'] += desc['fill_rows'] > 0
    print(json.dumps({'ok': True, 'cases': a.cases, 'threads': a.threads, 'seed': a.seed, 'files_checked': sum(n for n, _ in results),
                      'seen': seen, 'seconds': round(time.perf_counter() - t0, 1)}))
    return 0


if __name__ == '__main__':
    sys.exit(main())
