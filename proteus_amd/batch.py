"""Node-level batch driver: many HLS tiles (one runconfig each) over the GPUs of one node.

Tiles are independent, so the list of runconfigs is split statically and
contiguously (proteus_amd.shard.tile_range) and each GPU gets ONE worker process that
creates its HIP context once and runs its tiles back to back through
proteus_amd.dswx_hls.generate_dswx_layers.  Host scatter / gather only: no collective,
no inter-GPU traffic.  Workers report one JSON line per tile on stdout.

    python -m proteus_amd.batch --gpus 8 rc_000.yaml rc_001.yaml ...
"""
import argparse
import json
import os
import subprocess
import sys
import time

from . import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def plan(runconfigs, n_gpus, workers_per_gpu=1):
    """[(gpu, [runconfig, ...]), ...] -- contiguous, sizes differ by at most one.  With
    `workers_per_gpu` > 1 every GPU gets that many worker processes (the GPU part of a product run
    is milliseconds, the GeoTIFF codec on the host is what takes time: extra workers overlap it)."""
    out = []
    n = n_gpus * workers_per_gpu
    for w in range(n):
        lo, hi = shard.tile_range(len(runconfigs), w, n)
        out.append((w // workers_per_gpu, list(runconfigs[lo:hi])))
    return out


def _worker(device, runconfigs, skip_existing=False):
    """Runs inside the per-GPU process.  `skip_existing`: a tile whose requested output files all
    exist is not recomputed (tiles are idempotent: this is the driver's resume)."""
    import logging
    from . import dswx_hls as D
    logging.getLogger('dswx_hls').setLevel(logging.WARNING)
    D.get_context(device)                      # fail loudly before touching any tile
    rc = 0
    for path in runconfigs:
        t0 = time.perf_counter()
        try:
            args = D.get_dswx_hls_cli_parser().parse_args([path, '--device', str(device)])
            consts = D.parse_runconfig_file(path, args)
            kw = {k: getattr(args, k) for k in D.RunConfigConstants._FIELDS}
            for k in ('output_interpreted_band', 'output_rgb_file', 'output_infrared_rgb_file',
                      'output_binary_water', 'output_confidence_layer', 'output_diagnostic_layer',
                      'output_non_masked_dswx', 'output_shadow_masked_dswx', 'output_landcover',
                      'output_shadow_layer', 'output_cloud_layer', 'output_dem_layer',
                      'output_browse_image', 'scratch_dir', 'product_id', 'product_version',
                      'dem_file', 'dem_file_description', 'landcover_file',
                      'landcover_file_description', 'worldcover_file', 'worldcover_file_description',
                      'shoreline_shapefile', 'shoreline_shapefile_description',
                      'flag_offset_and_scale_inputs', 'landcover_mask', 'shadow_layer', 'ocean_mask'):
                kw[k] = getattr(args, k)
            wanted = [kw[k] for k in kw if k.startswith('output_') and kw[k]] + \
                ([args.output_file] if args.output_file else [])
            if skip_existing and wanted and all(os.path.exists(f) for f in wanted):
                print(json.dumps({'runconfig': path, 'device': device, 'ok': True, 'error': None,
                                  'skipped': True, 'seconds': 0.0}), flush=True)
                continue
            ok = D.generate_dswx_layers(args.input_list, args.output_file,
                                        hls_thresholds=consts.hls_thresholds, device=device, **kw)
            err = None
        except Exception as e:                  # one bad tile must not stop the slice
            ok, err = False, f'{type(e).__name__}: {e}'
        if not ok:
            rc = 1
        print(json.dumps({'runconfig': path, 'device': device, 'ok': bool(ok), 'error': err,
                          'seconds': round(time.perf_counter() - t0, 3)}), flush=True)
    return rc


def run_batch(runconfigs, n_gpus, python=sys.executable, workers_per_gpu=1, skip_existing=False):
    """Launch the workers; returns (all_ok, [per-tile result dicts in input order])."""
    procs = []
    for gpu, chunk in plan(runconfigs, n_gpus, workers_per_gpu):
        if not chunk:
            continue
        cmd = [python, '-m', 'proteus_amd.batch', '--worker', '--device', str(gpu)] + \
            (['--skip-existing'] if skip_existing else []) + chunk
        procs.append(subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    # every worker's pipes are drained concurrently: a worker with more result lines than a pipe holds (64 KiB, a few
    # hundred tiles) must not stall behind the worker the parent happens to be waiting for
    import threading
    outputs = [None] * len(procs)

    def drain(i):
        outputs[i] = procs[i].communicate()
    threads = [threading.Thread(target=drain, args=(i,)) for i in range(len(procs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    results, all_ok = {}, True
    for p, (out, err) in zip(procs, outputs):
        for line in out.splitlines():
            if line.startswith('{'):
                r = json.loads(line)
                results[r['runconfig']] = r
        if p.returncode != 0:
            all_ok = False
            sys.stderr.write(err[-2000:])
    ordered = [results.get(rc, {'runconfig': rc, 'ok': False, 'error': 'worker produced no result'})
               for rc in runconfigs]
    return all_ok and all(r['ok'] for r in ordered), ordered


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('runconfigs', nargs='+')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--workers-per-gpu', type=int, default=1,
                    help='worker processes per GPU (overlaps the host-side GeoTIFF codec of several tiles)')
    ap.add_argument('--skip-existing', action='store_true',
                    help='resume: do not recompute tiles whose requested output files all exist')
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--device', type=int, default=0, help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    if a.worker:
        return _worker(a.device, a.runconfigs, a.skip_existing)
    t0 = time.perf_counter()
    ok, results = run_batch(a.runconfigs, a.gpus, workers_per_gpu=max(1, a.workers_per_gpu),
                            skip_existing=a.skip_existing)
    for r in results:
        print(json.dumps(r))
    print(json.dumps({'tiles': len(results), 'gpus': a.gpus, 'ok': ok,
                      'seconds': round(time.perf_counter() - t0, 3)}))
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
