#!/usr/bin/env python3
"""Soak of the node-level driver's tiles-in-flight mode (proteus_amd.batch --in-flight): N synthetic tiles of mixed sizes and
sensors, some with ancillary inputs (DEM -> SHAD, land-cover maps -> LAND, ocean mask), through ONE worker process with K
tiles in flight on its threads -- one HIP context, the engine's lock, the pooled device and page-locked buffers, the codec's
shared thread pool -- every layer of every product against the numpy oracle computed from the arrays that were written.
Prints one JSON object; exit code 1 on the first mismatch.

    python tests/helpers/inflight_soak.py [--tiles 96] [--in-flight 8] [--workers 1] [--size 3660]

(Lives under tests/ because it uses the oracle; its name keeps pytest from collecting it.)"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_synthetic_hls as synth_hls                  # noqa: E402
from oracle import dswx_oracle as o                     # noqa: E402  (checker)
from proteus_amd import batch, geotiff                  # noqa: E402

LAYERS = (('B01_WTR', 'WTR'), ('B02_BWTR', 'BWTR'), ('B03_CONF', 'CONF'), ('B04_DIAG', 'DIAG'), ('B05_WTR-1', 'WTR-1'),
          ('B06_WTR-2', 'WTR-2'), ('B09_CLOUD', 'CLOUD'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tiles', type=int, default=96)
    ap.add_argument('--in-flight', type=int, default=8)
    ap.add_argument('--workers', type=int, default=1)
    ap.add_argument('--size', type=int, default=0, help='every tile this size (default: random 192 ... 487)')
    a = ap.parse_args()
    rng = np.random.default_rng(606)
    with tempfile.TemporaryDirectory() as d:
        rcs, tiles = [], []
        for t in range(a.tiles):
            size = a.size or int(rng.integers(24, 60)) * 8 + int(rng.integers(0, 8))
            anc = t % 5 == 0
            rc, _, _, s = synth_hls.make(os.path.join(d, f't{t}'), sensor=('L30', 'S30')[t % 2], size=size, tile=1000 + t,
                                         product_id=f'S{t}', ancillary=anc, ocean=anc, browse=t % 3 == 0)
            rcs.append(rc)
            tiles.append((size, anc, s))
        reports = []
        t0 = time.perf_counter()
        ok, res = batch.run_batch(rcs, 1, workers_per_gpu=a.workers, in_flight=a.in_flight, reports=reports)
        dt = time.perf_counter() - t0
        if not ok:
            print(json.dumps({'ok': False, 'why': 'a tile failed', 'results': [r for r in res if not r['ok']][:3]}))
            return 1
        checked = 0
        for t, (size, anc, s) in enumerate(tiles):
            out = os.path.join(d, f't{t}', 'output')
            kw = {}
            if anc:         # the layers the product itself made from the ancillary inputs: read back, then the chain must follow from them
                kw = dict(landcover=geotiff.read_geotiff(os.path.join(out, f'S{t}_v1.0_B07_LAND.tif'))[0],
                          shadow=geotiff.read_geotiff(os.path.join(out, f'S{t}_v1.0_B08_SHAD.tif'))[0],
                          ocean_mask=synth_hls.synth_tile(1000 + t, size, size, with_masks=True)['ocean'])
            exp = o.classify_tile(s['bands'], s['fmask'], **kw)
            for stem, layer in LAYERS:
                arr, _ = geotiff.read_geotiff(os.path.join(out, f'S{t}_v1.0_{stem}.tif'))
                if not np.array_equal(arr, exp[layer]):
                    print(json.dumps({'ok': False, 'tile': t, 'size': size, 'ancillary': anc, 'layer': layer,
                                      'wrong_pixels': int(np.count_nonzero(arr != exp[layer]))}))
                    return 1
                checked += 1
            if anc:         # the Float32 DEM layer and the first level of its CUBICSPLINE pyramid (made on the device)
                from proteus_amd.synth import synth_dem
                dem = synth_dem(1000 + t, size + 100, size + 100)[50:-50, 50:-50]
                got, _ = geotiff.read_geotiff(os.path.join(out, f'S{t}_v1.0_B10_DEM.tif'))
                ovr, _ = geotiff.read_geotiff(os.path.join(out, f'S{t}_v1.0_B10_DEM.tif'), overview=0)
                if not (np.array_equal(got, dem, equal_nan=True) and np.array_equal(ovr, geotiff.overview_cubicspline(dem, 4), equal_nan=True)):
                    print(json.dumps({'ok': False, 'tile': t, 'size': size, 'layer': 'DEM (image or overview)'}))
                    return 1
                checked += 1
            if t % 3 == 0:
                png = [f for f in os.listdir(out) if f.endswith('.png')]
                if len(png) != 1 or os.path.getsize(os.path.join(out, png[0])) < 100:
                    print(json.dumps({'ok': False, 'tile': t, 'why': 'browse PNG missing', 'files': sorted(os.listdir(out))}))
                    return 1
                checked += 1
        print(json.dumps({'ok': True, 'tiles': a.tiles, 'in_flight': a.in_flight, 'workers': a.workers, 'layers_checked': checked,
                          'with_ancillary_inputs': sum(1 for _, anc, _ in tiles if anc), 'seconds': round(dt, 2),
                          'worker_reports': [{k: r[k] for k in ('tiles', 'in_flight', 'bring_up_s', 'tiles_s')} for r in reports]}))
    return 0


if __name__ == '__main__':
    sys.exit(main())
