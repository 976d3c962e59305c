#!/usr/bin/env python3
"""Product throughput of the node-level driver on ONE GPU: N full-size synthetic tiles (GeoTIFFs in,
cloud-optimized layers out) with 1 and with several worker processes per GPU.  Prints one JSON object.

    batch_throughput.py [N] [SIZE] [--scene]     --scene: spatially coherent scenes (make_synthetic_hls.scene_tile; four
                                                 distinct ones, repeated) instead of the per-pixel recipe whose class maps are noise"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_synthetic_hls as synth_hls          # noqa: E402
from proteus_amd import batch                   # noqa: E402


def main():
    scene = '--scene' in sys.argv
    fill = float(sys.argv[sys.argv.index('--fill-rows') + 1]) if '--fill-rows' in sys.argv else 0.0      # nodata rows at the top of every tile
    pos = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith('--') and sys.argv[i - 1] not in ('--grid', '--fill-rows')]
    n = int(pos[0]) if len(pos) > 0 else 8
    size = int(pos[1]) if len(pos) > 1 else 3660
    out = {'tiles': n, 'size': size, 'nodata_rows': fill, 'inputs': 'coherent scenes (4 distinct)' if scene else 'per-pixel recipe (noise-like class maps)'}
    with tempfile.TemporaryDirectory() as d:
        rcs = [synth_hls.make(os.path.join(d, f't{i}'), sensor=('L30', 'S30')[i % 2], size=size, tile=i % 4 if scene else i,
                              product_id=f'P{i}', scene=scene, fill_rows=fill)[0] for i in range(n)]
        import shutil
        grid = ((1, 1), (1, 3), (2, 2)) if scene else ((1, 1), (1, 3), (2, 2), (4, 1), (8, 1))
        if '--grid' in sys.argv:            # workers per GPU x tiles in flight, e.g. --grid 1x3,2x2,2x3
            grid = tuple(tuple(int(v) for v in g.split('x')) for g in sys.argv[sys.argv.index('--grid') + 1].split(','))
        for wpg, in_flight in grid:
            for i in range(n):
                shutil.rmtree(os.path.join(d, f't{i}', 'output'), ignore_errors=True)
            reports = []
            t0 = time.perf_counter()
            ok, res = batch.run_batch(rcs, 1, workers_per_gpu=wpg, in_flight=in_flight, reports=reports)
            dt = time.perf_counter() - t0
            assert ok, res
            tiles_s = max(r['tiles_s'] for r in reports)
            rec = {'seconds': round(dt, 2), 'tiles_per_s': round(n / dt, 2), 'Mpix_per_s': round(n * size * size / dt / 1e6, 1),
                   # without the process start + HIP bring-up of the workers (a long-lived service pays it once)
                   'bring_up_s': max(r['bring_up_s'] for r in reports), 'tiles_s': tiles_s,
                   'steady_tiles_per_s': round(n / tiles_s, 2),
                   'cpu_core_s_per_tile': round(sum(r.get('cpu_s', 0.0) for r in reports) / n, 3),
                   'tile_seconds_median': sorted(r['seconds'] for r in res)[len(res) // 2]}
            if (wpg, in_flight) in ((1, 1), (1, 3)):
                rec['stages'] = reports[0]['stages']
            out[f'workers_per_gpu_{wpg}_in_flight_{in_flight}'] = rec
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
