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


def _one_tile(D, device, path, skip_existing):
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
            return {'runconfig': path, 'device': device, 'ok': True, 'error': None, 'skipped': True, 'seconds': 0.0}
        ok = D.generate_dswx_layers(args.input_list, args.output_file,
                                    hls_thresholds=consts.hls_thresholds, device=device, **kw)
        err = None
    except Exception as e:                  # one bad tile must not stop the slice
        ok, err = False, f'{type(e).__name__}: {e}'
    return {'runconfig': path, 'device': device, 'ok': bool(ok), 'error': err,
            'seconds': round(time.perf_counter() - t0, 3)}


def _worker(device, runconfigs, skip_existing=False, in_flight=3, with_stages=False):
    """Runs inside the per-GPU process.  `skip_existing`: a tile whose requested output files all
    exist is not recomputed (tiles are idempotent: this is the driver's resume).

    `in_flight` tiles are processed side by side on threads of THIS process: the GPU part of a tile is milliseconds
    (and serialised by the engine's lock), its host part -- inflating the band files, deflating the layers, on native
    threads without the interpreter lock -- is what takes time, so tile k + 1 is being read and tile k - 1 written
    while tile k is on the device, with ONE HIP context per GPU."""
    import logging
    import threading
    from concurrent.futures import ThreadPoolExecutor
    t_start = time.perf_counter()
    from . import dswx_hls as D
    from . import stages
    logging.getLogger('dswx_hls').setLevel(logging.WARNING)
    D.get_context(device)                      # fail loudly before touching any tile
    t_ready = time.perf_counter()
    cpu0 = os.times()
    if with_stages:
        stages.start()
    out_lock = threading.Lock()
    rc = [0]

    def run(path):
        r = _one_tile(D, device, path, skip_existing)
        with out_lock:
            if not r['ok']:
                rc[0] = 1
            print(json.dumps(r), flush=True)

    n = max(1, min(int(in_flight), len(runconfigs)))
    if n == 1:
        for path in runconfigs:
            run(path)
    else:
        with ThreadPoolExecutor(n, thread_name_prefix='dswx-tile') as ex:
            list(ex.map(run, runconfigs))
    if with_stages:
        print(json.dumps({'worker_report': {'device': device, 'tiles': len(runconfigs), 'in_flight': n,
                                            'bring_up_s': round(t_ready - t_start, 3),
                                            'tiles_s': round(time.perf_counter() - t_ready, 3),
                                            # processor time of this process (all its threads, the codec's included)
                                            'cpu_s': round(sum(os.times()[:2]) - sum(cpu0[:2]), 3),
                                            'stages': stages.stop()}}), flush=True)
    return rc[0]


def run_batch(runconfigs, n_gpus, python=sys.executable, workers_per_gpu=1, skip_existing=False, in_flight=3,
              reports=None):
    """Launch the workers; returns (all_ok, [per-tile result dicts in input order]).  `reports`: a list that receives
    every worker's stage report (proteus_amd.stages: where its wall time went)."""
    procs = []
    jobs = [(gpu, chunk) for gpu, chunk in plan(runconfigs, n_gpus, workers_per_gpu) if chunk]
    # the host codec is what a product run waits for, and the workers share the machine's processors (or the container's
    # CPU quota): every worker gets its share, so that N workers do not start N pools of the full size and throttle each
    # other (measured on a 16-core quota: 8 workers with full-size pools 6.0 tiles/s, one worker 7.2)
    env = dict(os.environ)
    if len(jobs) > 1 and 'DSWX_CPU_SHARE' not in env:
        try:
            from . import codec
            env['DSWX_CPU_SHARE'] = str(max(2, codec.cpu_budget() // len(jobs)))
        except Exception:           # noqa: BLE001  (no native codec: the zlib module's pool is small anyway)
            pass
    for gpu, chunk in jobs:
        cmd = [python, '-m', 'proteus_amd.batch', '--worker', '--device', str(gpu), '--in-flight', str(in_flight)] + \
            (['--skip-existing'] if skip_existing else []) + (['--stages'] if reports is not None else []) + chunk
        procs.append(subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=env))
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
            if line.startswith('{"worker_report"'):
                if reports is not None:
                    reports.append(json.loads(line)['worker_report'])
            elif line.startswith('{'):
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
    ap.add_argument('--in-flight', type=int, default=3,
                    help='tiles processed side by side inside one worker (read k+1 / device k / write k-1 overlap)')
    ap.add_argument('--stages', action='store_true', help='workers report where their wall time went (proteus_amd.stages)')
    ap.add_argument('--skip-existing', action='store_true',
                    help='resume: do not recompute tiles whose requested output files all exist')
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--device', type=int, default=0, help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    if a.worker:
        return _worker(a.device, a.runconfigs, a.skip_existing, a.in_flight, a.stages)
    t0 = time.perf_counter()
    reports = [] if a.stages else None
    ok, results = run_batch(a.runconfigs, a.gpus, workers_per_gpu=max(1, a.workers_per_gpu),
                            skip_existing=a.skip_existing, in_flight=max(1, a.in_flight), reports=reports)
    for r in results:
        print(json.dumps(r))
    for r in reports or ():
        print(json.dumps({'worker_report': r}))
    print(json.dumps({'tiles': len(results), 'gpus': a.gpus, 'ok': ok,
                      'seconds': round(time.perf_counter() - t0, 3)}))
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
