#!/usr/bin/env python3
"""Benchmark of the DSWx-HLS per-pixel hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--tiles T] [--masks]

One "step" = one pass of the fused classify kernel over a device-resident batch of
T synthetic 3660x3660 HLS.L30 tiles (6 int16 bands + Fmask; `--masks` adds
LAND/SHAD/OCEAN) per GPU.  Inputs are generated in HBM before the timed region.
For N > 1 launch with torch.distributed.run (one rank per GPU); tiles are sharded
by rank, there is no data-path collective (weak scaling: T tiles per GPU).

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TILE = 3660


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--tiles', type=int, default=256, help='tiles per GPU per step')
    ap.add_argument('--masks', action='store_true',
                    help='also stream LAND/SHAD/OCEAN planes (BASELINE config 5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-single-tile', action='store_true',
                    help='skip the configs[1] leg (profiling runs: keeps the kernel statistics to the batch launches)')
    ap.add_argument('--cpu-parallel-worker', type=int, default=0, help=argparse.SUPPRESS)
    return ap.parse_args()


def cpu_baseline_sample(n_tiles=4):
    """The numpy restatement of the reference path (oracle, kind 'port') on a bounded sample of
    the workload: `n_tiles` synthetic 3660x3660 tiles (~10 s), single thread as the reference runs."""
    import numpy as np
    from oracle import dswx_oracle as o
    from proteus_amd.synth import synth_tile
    dt = 0.0
    for t in range(n_tiles):
        s = synth_tile(t, TILE, TILE)
        t0 = time.perf_counter()
        o.classify_tile(s['bands'], s['fmask'])
        dt += time.perf_counter() - t0
    return {'value': round(n_tiles * TILE * TILE / dt / 1e6, 3), 'unit': 'Mpixels/s', 'cores': 1,
            'kind': 'port',
            'sample': f'{n_tiles} synthetic {TILE}x{TILE} L30 tiles, numpy {np.__version__} '
                      f'oracle/dswx_oracle.py classify_tile, {dt:.2f} s, '
                      f'host has {os.cpu_count()} logical cores'}


def cpu_baseline_next_rows(dem, worldcover_up3, copernicus, forest_classes, sample=1500):
    """CPU-baseline leg of tools/next_rows_bench.py (kept here because only bench.py's CPU baseline may
    use the oracle outside tests/): the numpy oracle's shadow layer, LAND aggregation and 'cover'-mode
    chain on a sample x sample window, one core, scaled by area to a 3660 x 3660 tile."""
    from oracle import dswx_oracle as o
    from proteus_amd.synth import synth_tile
    S = sample
    scale = (TILE * TILE) / (S * S)
    t0 = time.perf_counter()
    o.compute_opera_shadow_layer(dem[:S + 100, :S + 100], 141.0, 35.0, -5.0, 40.0)
    t_shadow = (time.perf_counter() - t0) * scale
    t0 = time.perf_counter()
    o.landcover_mask_from_warped(worldcover_up3[:3 * S, :3 * S], copernicus[:S, :S], forest_classes)
    t_land = (time.perf_counter() - t0) * scale
    s = synth_tile(0, S, S, with_masks=True)
    t0 = time.perf_counter()
    o.classify_tile(s['bands'], s['fmask'], landcover=s['land'], shadow=s['shad'], ocean_mask=s['ocean'],
                    mask_adjacent_to_cloud_mode='cover')
    t_cover = (time.perf_counter() - t0) * scale
    return {'shadow_s_per_tile': t_shadow, 'landcover_s_per_tile': t_land, 'cover_s_per_tile': t_cover,
            'note': f'numpy oracle on one core, {S}x{S} sample scaled by area to {TILE}x{TILE}'}


_WORKER_TILE = None


def _cpu_worker_prepare(tile):
    global _WORKER_TILE
    from proteus_amd.synth import synth_tile
    _WORKER_TILE = synth_tile(tile, TILE, TILE)
    return tile


def _cpu_worker_classify(_):
    from oracle import dswx_oracle as o
    t0 = time.perf_counter()
    o.classify_tile(_WORKER_TILE['bands'], _WORKER_TILE['fmask'])
    return time.perf_counter() - t0


def cpu_parallel_main(workers):
    """Child-process mode (never touches the GPU): `workers` processes, one synthetic tile
    each, classified concurrently by the numpy oracle; prints one JSON object."""
    import multiprocessing as mp
    with mp.Pool(workers) as pool:
        pool.map(_cpu_worker_prepare, range(workers), chunksize=1)
        t0 = time.perf_counter()
        per = pool.map(_cpu_worker_classify, range(workers), chunksize=1)
        dt = time.perf_counter() - t0
    print(json.dumps({'value': round(workers * TILE * TILE / dt / 1e6, 3), 'unit': 'Mpixels/s',
                      'cores': workers,
                      'sample': f'{workers} worker processes x 1 synthetic tile each, concurrently, '
                                f'{dt:.2f} s wall (slowest worker {max(per):.2f} s)'}))


def cpu_baseline_parallel():
    """tile-parallel numpy oracle on min(cores, 32) worker processes (bounded by
    free memory, ~4 GB per worker), run in a child process that never initialises the GPU."""
    import subprocess
    workers = min(os.cpu_count() or 1, 32)
    try:
        import psutil
        workers = max(1, min(workers, int(psutil.virtual_memory().available // (4 << 30))))
    except ImportError:
        workers = min(workers, 8)
    if workers < 2:
        return None
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-parallel-worker', str(workers)],
                           capture_output=True, text=True, timeout=300)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:      # the baseline is a report, never a reason to lose the bench line
        return {'error': str(e)[:200]}


def single_tile_leg(ctx, params, masks, reps=50):
    """BASELINE.json configs[1] beside the headline: ONE device-resident 3660 x 3660 tile classified
    repeatedly.  Its 281 MB working set is partly served by the 256 MiB Infinity Cache and a launch
    lasts < 0.1 ms, so this is a latency figure, not an HBM figure."""
    from proteus_amd import _capi
    from proteus_amd.synth import SEED
    b1 = _capi.DeviceBatch(ctx, 1, TILE, TILE, masks=masks)
    b1.synth(SEED, tile0=7)
    for _ in range(5):
        b1.classify(params)
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        b1.classify(params)
    ctx.record(e1)
    ctx.synchronize()
    ms = ctx.elapsed_ms(e0, e1) / reps
    ctx.destroy_event(e0)
    ctx.destroy_event(e1)
    b1.free()
    return {'value': round(TILE * TILE / ms / 1e3, 1), 'unit': 'Mpixels/s', 'ms_per_launch': round(ms, 4),
            'note': 'BASELINE configs[1]: one resident tile, back-to-back launches incl. the counters kernel; '
                    'working set 281 MB (Infinity-Cache assisted), launch-latency bound'}


def parity_spot_check(ctx, batch, params, tile):
    """Not timed: one tile of the batch against the scalar C oracle."""
    import numpy as np
    from oracle import c_oracle
    from proteus_amd import _capi
    bands = [batch.read_tile(b, tile) for b in _capi.BAND_NAMES]
    kw = {}
    if batch.masks:
        kw = {m: batch.read_tile(m, tile) for m in ('land', 'shad', 'ocean')}
    exp = c_oracle.classify(params, bands, batch.read_tile('fmask', tile), **kw)
    for key in ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
        if not np.array_equal(batch.read_tile(key, tile), exp[key]):
            return f'MISMATCH in {key}'
    if batch.read_counters()[tile].tolist() != exp['counters'].tolist():
        return 'MISMATCH in counters'
    return 'bit-exact'


def main():
    args = parse_args()
    if args.cpu_parallel_worker:
        cpu_parallel_main(args.cpu_parallel_worker)
        return
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('for --gpus N > 1 launch with: python -m torch.distributed.run '
                             '--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py ...')
        raise SystemExit(f'WORLD_SIZE={world} does not match --gpus {args.gpus}')

    import torch
    from proteus_amd import build as _build
    if not os.path.exists(_build.LIB_PATH):     # fresh checkout: compile the HIP library (never a CPU fallback)
        if local_rank == 0:
            _build.build()                      # os.replace at the end: the file appears complete or not at all
        else:
            for _ in range(600):
                if os.path.exists(_build.LIB_PATH):
                    break
                time.sleep(0.5)
    from proteus_amd import _capi, shard
    from proteus_amd.synth import SEED

    if os.environ.get('DSWX_BENCH_SHARE_DEVICE') == '1':
        # functional test of the N > 1 code path on a 1-GPU box: every rank on device 0, gloo as
        # the control plane (RCCL refuses two ranks on one GPU).  Not a measurement.
        local_rank = 0
        torch.cuda.set_device(0)
        cp = shard.ControlPlane(backend='gloo', device=None)
    else:
        torch.cuda.set_device(local_rank)
        cp = shard.ControlPlane(backend='nccl', device=torch.device('cuda', local_rank))

    ctx = _capi.Context(local_rank)        # raises if the HIP extension / GPU is missing
    params = _capi.default_params()
    n_tiles = args.tiles
    batch = _capi.DeviceBatch(ctx, n_tiles, TILE, TILE, masks=args.masks)
    tile0, _ = shard.weak_tile_range(n_tiles, rank)     # rank r owns tiles [r*T, (r+1)*T)
    batch.synth(SEED, tile0=tile0)
    ctx.synchronize()
    barrier = cp.barrier

    for _ in range(args.warmup):
        batch.classify(params)
    ctx.synchronize()
    kernel_info = ctx.last_kernel_info()

    starts = [ctx.event() for _ in range(args.steps)]
    stops = [ctx.event() for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ctx.record(starts[k])
        batch.classify(params)
        ctx.record(stops[k])
    ctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    launch_ms = [ctx.elapsed_ms(a, b) for a, b in zip(starts, stops)]
    for e in starts + stops:
        ctx.destroy_event(e)
    elapsed = cp.max_over_ranks(elapsed)

    parity = None
    if rank == 0 and not args.no_parity:
        try:
            parity = parity_spot_check(ctx, batch, params, n_tiles // 2)
        except Exception as e:          # the checker failing is reported, not fatal to the measurement
            parity = f'not checked ({type(e).__name__}: {e})'[:300]

    if rank == 0:
        px_per_launch = n_tiles * TILE * TILE
        bytes_per_px = 24 if args.masks else 21      # 13+8 (16+8 with masks)
        avg_ms = sum(launch_ms) / len(launch_ms)
        achieved = px_per_launch * bytes_per_px / (avg_ms * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        pmc_note = None
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            key = 'masks' if args.masks else 'plain'
            if key in pmc and pmc[key].get('tiles') == n_tiles:
                traffic = pmc[key]['hbm_bytes_per_launch']
                pmc_note = pmc[key].get('source')
        out = {
            'metric': 'Mpixels/sec DSWx classify (3660^2 7-band HLS tiles)',
            'value': round(world * px_per_launch * args.steps / elapsed / 1e6, 1),
            'unit': 'Mpixels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'int16+f64', 'data': 'synthetic',
            'config': {'workload': f'{n_tiles} synthetic {TILE}x{TILE} HLS.L30 tiles per GPU '
                                   f'per step, device-resident band-planar batch '
                                   f'(tile stride {batch.tile_stride} px = 256-byte aligned tile starts)'
                                   + (', LAND+SHAD+OCEAN planes' if args.masks else ''),
                       'tiles_per_gpu': n_tiles, 'tile': [TILE, TILE], 'tile_stride_px': batch.tile_stride,
                       'planes_in': 10 if args.masks else 7, 'planes_out': 7,
                       'sharding': f'tiles by rank x{world}, no collective',
                       'control_plane': cp.backend,
                       'kernel': kernel_info},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': traffic,
                         'algorithmic_bytes_per_pixel': bytes_per_px,
                         'pixels_per_launch': px_per_launch,
                         'launch_ms_avg': round(avg_ms, 4), 'launch_ms_min': round(min(launch_ms), 4),
                         'read_frac_of_peak': round(achieved * (bytes_per_px - 8) / bytes_per_px
                                                    / HBM_PEAK_GBS, 4),
                         'traffic_source': pmc_note},
            'parity_check': parity,
        }
        if world == 1 and not args.no_single_tile:
            try:
                out['single_tile'] = single_tile_leg(ctx, params, args.masks)
            except Exception as e:
                out['single_tile'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline_sample()
                par = cpu_baseline_parallel()
                if par:
                    out['cpu_baseline']['tile_parallel'] = par
            except Exception as e:      # a reported baseline must never cost the bench line
                out['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        print(json.dumps(out), flush=True)
    batch.free()
    ctx.close()
    cp.close()


if __name__ == '__main__':
    main()
