#!/usr/bin/env python3
"""Benchmark of the DSWx-HLS per-pixel hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--tiles T] [--total-tiles M] [--masks] [--chain] [--placement ...]

One "step" = one pass of the fused classify kernel over the tiles a GPU owns: synthetic
3660x3660 HLS.L30 tiles (6 int16 bands + Fmask; `--masks` adds LAND/SHAD/OCEAN), generated
in HBM before the timed region.
  default (weak scaling)        every GPU owns T = 256 resident tiles: BASELINE configs[2] at N = 1
  --total-tiles M (strong)      BASELINE configs[3]: M (e.g. 4096) tiles split contiguously over the
                                N ranks (proteus_amd.shard.tile_range); a rank walks its share in
                                resident chunks of <= T tiles (default 512), so N = 1 works too
  --chain                       BASELINE configs[4]'s per-pixel chain per GPU: a step = terrain shadow layer from
                                DEMs + LAND aggregation from WorldCover / CGLS, written straight into the batch's
                                SHAD / LAND planes, + the classifier with SHAD + LAND + OCEAN
The resident batch is allocated by the library (dswx_batch_create) and PLACED before warm-up (--placement, DESIGN.md
sections 5 - 6): by default dswx_batch_place_slide times ~100 placements of the output planes inside a VMM-backed
address range 48 GiB longer than they are and keeps the best; `roofline.frac_first_come_placement`,
`roofline.frac_kept_placement_probe` and `roofline.realloc_spread` keep what unplaced allocations give in the same line.
Every rank checks tiles of its own batch against the oracle after timing (`parity_check.ranks`).
Tiles are independent: there is no data-path collective.  RCCL (torch.distributed 'nccl') carries
only the barrier around the timed region and the MAX over ranks of the elapsed time.

`python bench.py --gpus N` with N > 1 and no torchrun environment launches its own N ranks
(torch.distributed.run as a child process, BEFORE this process touches the GPU) and relays
rank 0's line.  Under torchrun (RANK / WORLD_SIZE set) it is a rank.

Prints ONE JSON line on rank 0 (see DESIGN.md section 6 for every field).
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
STRONG_TOTAL_TILES = 4096          # BASELINE configs[3]
STRONG_CHUNK_TILES = 512           # its per-GPU share at 8 GPUs, resident at once (89 GB + 55 GB)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--tiles', type=int, default=0,
                    help='resident tiles per GPU (weak scaling: the work of a step, default 256 = BASELINE configs[2]; '
                         'strong: the chunk size, default 512 = the per-GPU share of configs[3] at 8 GPUs)')
    ap.add_argument('--total-tiles', type=int, default=0,
                    help='strong scaling: tiles of the whole job, split over the ranks (BASELINE configs[3]: 4096)')
    ap.add_argument('--require-rccl', action='store_true',
                    help='fail (every rank; the line carries no number, only the reason) if the RCCL control plane cannot come up on every rank.  Default: fall '
                         'back to gloo LOUDLY -- `config.control_plane` says why and `rccl_ranks` is 0; RCCL carries only the '
                         'barriers and the MAX / SUM of a few scalars here, tiles are independent')
    ap.add_argument('--allow-gloo', action='store_true', help=argparse.SUPPRESS)       # r04 spelling: now the default
    ap.add_argument('--preflight', action='store_true',
                    help='bring the ranks up, print the preflight record (per rank: device PCI address + UUID, free HBM vs '
                         'what the resident chunk needs, control-plane round trip) as JSON and exit.  The same record is '
                         'taken before every N > 1 measurement and stored in the line as `preflight`')
    ap.add_argument('--realloc-repeats', type=int, default=5,
                    help='N = 1: after the timed region, re-allocate the batch this many times and report the '
                         'spread of the kernel rate (it depends on where the arena lands, DESIGN.md section 5)')
    ap.add_argument('--placement', default='slide', choices=['slide', 'search', 'first', 'arena'],
                    help='how the resident batch is placed before warm-up (DESIGN.md section 6): slide = '
                         'dswx_batch_place_slide (the packed output region timed at 25 offsets of a range 48 GiB longer than '
                         'itself, then every plane refined among the free places of that range), search = dswx_batch_place_search (--placement-trials candidate allocations per output '
                         'plane), first = one allocation per output plane as they come, arena = all planes in one hipMalloc')
    ap.add_argument('--slide-refine', type=int, default=1, help='--placement slide: passes of per-plane refinement (0 = the packed region only)')
    ap.add_argument('--slide-slack-gib', type=float, default=48.0, help='--placement slide: length of the range beyond the planes')
    ap.add_argument('--placement-trials', type=int, default=None,
                    help='implies --placement search (>= 2 candidates per output plane), first (1) or arena (0)')
    ap.add_argument('--plan-only', action='store_true',
                    help='no GPU work: bring the ranks up (gloo), print the sharding plan of this command line as '
                         'JSON and exit (tests/test_shard_gloo.py drives the launcher path with it)')
    ap.add_argument('--masks', action='store_true',
                    help='also stream LAND/SHAD/OCEAN planes (BASELINE config 5)')
    ap.add_argument('--chain', action='store_true',
                    help="BASELINE configs[4]'s per-pixel chain, device-resident: every step computes the terrain shadow "
                         'layer from DEMs and the LAND layer from WorldCover + CGLS maps straight into the SHAD / LAND planes of '
                         'the batch (dswx_shadow_layer_batch, dswx_landcover_mask_batch) and classifies with SHAD + LAND + '
                         'OCEAN on; weak scaling only')
    ap.add_argument('--scaled', action='store_true',
                    help="the reference's flag_offset_and_scale_inputs (dswx_hls.py:2300-2302): the chain on float32 "
                         'reflectances 0.0001 * x, thresholds in those units -- the float32 instantiation of the fused kernel')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--distinct-chunks', action='store_true',
                    help='strong scaling: after the timed region, generate every later chunk of the rank\'s share with '
                         'its own tile indices, classify it and check its first and last tile (the timed walk re-streams '
                         'the resident planes)')
    ap.add_argument('--no-host-path', action='store_true',
                    help='skip the end-to-end leg (dswx_classify_host on 4 tiles from page-locked and from pageable planes, '
                         'all ranks at once; after the timed region)')
    ap.add_argument('--no-strong', action='store_true',
                    help='--gpus N > 1 without --tiles / --total-tiles measures the weak record AND BASELINE configs[3] '
                         '(4096 tiles over the ranks, the `strong` sub-record of the line); this keeps the weak record only')
    ap.add_argument('--also-strong', action='store_true',
                    help='measure the `strong` sub-record (BASELINE configs[3]) after the first record whatever N is: at N = 1 '
                         'the one rank walks all 4096 tiles in 512-tile chunks (the weak batch is freed first: the same '
                         'sequence an N > 1 rank goes through, at its real sizes)')
    ap.add_argument('--no-product-run', action='store_true',
                    help='skip the product-run leg of the plain N = 1 command (one full GeoTIFF-in / COG-out product of a '
                         '3660 x 3660 tile in a child process, after the timed regions)')
    ap.add_argument('--no-single-tile', action='store_true',
                    help='skip the configs[1] leg (profiling runs: keeps the kernel statistics to the batch launches)')
    ap.add_argument('--cpu-parallel-worker', type=int, default=0, help=argparse.SUPPRESS)
    # the two-record plain command at toy sizes (tests/test_gpu_multirank.py: two ranks share one device there)
    ap.add_argument('--plain-tiles', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--strong-total', type=int, default=STRONG_TOTAL_TILES, help=argparse.SUPPRESS)
    ap.add_argument('--strong-chunk', type=int, default=STRONG_CHUNK_TILES, help=argparse.SUPPRESS)
    # tests of the control flow (tests/test_gpu_multirank.py) shrink the tile: the CPU-side checker dominates their time
    ap.add_argument('--tile-size', type=int, default=TILE, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.tile_size != TILE:
        globals()['TILE'] = args.tile_size
    # "plain": the command line names no workload -- what the driver runs (`bench.py --gpus N --steps K --warmup W`)
    args.plain_command = args.tiles <= 0 and args.total_tiles <= 0 and not args.chain and not args.masks and not args.scaled
    if args.chain:
        if args.total_tiles > 0:
            ap.error('--chain is a weak-scaling mode (one resident batch per GPU)')
        args.masks = True
    if args.tiles <= 0:
        args.tiles = 512 if args.total_tiles > 0 else (args.plain_tiles or 256)
    if args.placement_trials is not None:
        args.placement = 'arena' if args.placement_trials <= 0 else 'first' if args.placement_trials == 1 else 'search'
    elif args.placement == 'search':
        args.placement_trials = 6
    return args


SCALED_UNITS = 1e-4            # --scaled: HLS scale_factor; the reflectance thresholds below are the defaults x 1e-4
SCALED_THRESHOLDS = dict(wigt=0.124, awgt=0.0, pswt_1_mndwi=-0.44, pswt_1_nir=0.15, pswt_1_swir1=0.09, pswt_1_ndvi=0.7,
                         pswt_2_mndwi=-0.5, pswt_2_blue=0.1, pswt_2_nir=0.25, pswt_2_swir1=0.3, pswt_2_swir2=0.1,
                         lcmask_nir=0.12)


def bench_params(args):
    from proteus_amd import _capi
    if not args.scaled:
        return _capi.default_params()
    return _capi.make_params(SCALED_THRESHOLDS, offset_and_scale=[(SCALED_UNITS, 0.0)] * 6, aerosol_max_nir=1000 * SCALED_UNITS)


def cpu_quota_cores():
    """Processors this process may really use: the logical cores, cut down to the container's CPU bandwidth quota (cgroup
    v2 cpu.max / v1 cfs quota).  The GPU boxes of this project show 256 logical cores under a quota of 16: worker pools
    sized by os.cpu_count() are throttled, and a `cores` figure that ignores the quota overstates the baseline's hardware."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota, period = None, 100000
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        quota, period = (None if q == 'max' else int(q)), int(p)
    except (OSError, ValueError):
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        except (OSError, ValueError):
            quota = None
    if quota and quota > 0 and period > 0:
        n = max(1, min(n, -(-quota // period)))
    return n


def cpu_baseline_sample(n_tiles=4):
    """The numpy restatement of the reference path (oracle, kind 'port') on a bounded sample of
    the workload: `n_tiles` synthetic 3660x3660 tiles (~10 s), single thread as the reference runs.
    The port makes the same whole-array passes as the reference's functions; timed stage by stage
    against them in the build container it takes 0.975x their time (3.94 s vs 4.04 s per tile,
    profiles/r02_cpu_port_vs_reference.json), so this figure is the reference's own speed to ~3 %."""
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
                      f'oracle/dswx_oracle.py classify_tile (same array passes as the reference functions: '
                      f'0.975x their time stage by stage, profiles/r02_cpu_port_vs_reference.json), '
                      f'{dt:.2f} s, host has {os.cpu_count()} logical cores, CPU quota of this container {cpu_quota_cores()} cores'}


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
    free memory, ~4 GB per worker; cores = what the container's CPU quota allows, cpu_quota_cores), run in a child
    process that never initialises the GPU."""
    import subprocess
    workers = min(cpu_quota_cores(), 32)
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


def product_run_leg(timeout_s=240):
    """The number a user of `dswx_hls.py <runconfig>` sees, beside the kernel's: ONE full product run of a 3660 x 3660
    tile -- seven DEFLATE GeoTIFFs in, seven cloud-optimized layers out, through proteus_amd.dswx_hls.generate_dswx_layers
    (inflate on host threads, untile / classify / COG blocks + overviews on the device, deflate on host threads) -- in a
    CHILD process with its own HIP context (tools/e2e_time.py: best of three), on the bench's synthetic recipe (a surface
    type per pixel: noise-like class maps, DEFLATE at its slowest) and on a spatially coherent scene.  After the timed
    regions; never `value`; a failure is a record."""
    import subprocess
    if 'rocprof' in os.environ.get('LD_PRELOAD', '') or any(k.startswith(('ROCPROF', 'ROCPROFILER')) for k in os.environ):
        return {'skipped': 'running under a profiler: its trace stays the bench kernels\' (run `python tools/e2e_time.py` on its own)'}
    out = {}
    for key, extra in (('synthetic_recipe_tile', []), ('coherent_scene_tile', ['--scene'])):
        try:
            r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'e2e_time.py'),
                                str(TILE)] + extra, capture_output=True, text=True, timeout=timeout_s)
            d = json.loads(r.stdout[r.stdout.index('{'):])
            out[key] = {'seconds_per_product': d['generate_dswx_layers_s_io_threads_default'],
                        'Mpixels_per_s': round(TILE * TILE / d['generate_dswx_layers_s_io_threads_default'] / 1e6, 1),
                        'codec_threads': d['io_threads_default'],
                        'seconds_on_one_codec_thread': d['generate_dswx_layers_s_io_threads_1'],
                        'input_MB': d.get('input_MB'), 'output_MB': d.get('output_MB'),
                        'stages_wall_s': {n: v['wall_s'] for n, v in d['stages_io_threads_default']['stages'].items()}}
        except Exception as e:      # noqa: BLE001  (a reported figure must never cost the bench line)
            out[key] = {'error': f'{type(e).__name__}: {e}'[:300]}
    out['note'] = ('one fresh child process per input kind, warm product runs inside it (best of three); host codec = '
                   'libdswx_codec.so on the container\'s CPU quota; DESIGN.md section 6, profiles/r06_product_run.json')
    return out


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


def parity_spot_check(ctx, batch, params, tiles, tile0=0):
    """Not timed: tiles of the resident batch against the scalar C oracle, and their INPUT planes against the
    numpy generator for the tile index the plan gives them (`tile0` + position: a rank whose share starts at
    tile `lo` must hold tiles lo, lo+1, ... -- a wrong offset is a parity failure, not a detail)."""
    import numpy as np
    from oracle import c_oracle
    from proteus_amd import _capi
    from proteus_amd.synth import synth_tile
    cnt = batch.read_counters()
    checked = [tile0 + t for t in tiles]
    for tile in tiles:
        bands = [batch.read_tile(b, tile) for b in _capi.BAND_NAMES]
        fmask = batch.read_tile('fmask', tile)
        kw = {}
        if batch.masks:
            kw = {m: batch.read_tile(m, tile) for m in ('land', 'shad', 'ocean')}
        want = synth_tile(tile0 + tile, TILE, TILE, with_masks=batch.masks)
        same = all(np.array_equal(a, b) for a, b in zip(bands, want['bands'])) and np.array_equal(fmask, want['fmask']) \
            and all(np.array_equal(kw[m], want[m]) for m in kw)
        if not same:
            return {'tiles': checked, 'result': f'resident tile {tile} is not synthetic tile {tile0 + tile}'}
        exp = c_oracle.classify(params, bands, fmask, **kw)
        for key in ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
            if not np.array_equal(batch.read_tile(key, tile), exp[key]):
                return {'tiles': checked, 'result': f'MISMATCH in {key} of tile {tile0 + tile}'}
        if cnt[tile].tolist() != exp['counters'].tolist():
            return {'tiles': checked, 'result': f'MISMATCH in counters of tile {tile0 + tile}'}
    return {'tiles': checked, 'result': 'bit-exact'}


def rank_parity(ctx, batch, params, rank, tile0, n_tiles, chunks, distinct):
    """This rank's parity record (every rank runs it; the records are gathered into the line): first, middle and
    LAST tile of the resident batch (at 256 tiles the last sits past 2^31 pixels / 2^32 bytes into every plane);
    with --distinct-chunks every later chunk of a strong-scaling walk is generated with ITS tile indices, classified
    and checked on its first and last tile, so every chunk of the rank's share is classified once under a check."""
    from proteus_amd.synth import SEED
    if len(chunks) > 1:                 # leave the full resident batch classified
        batch.classify(params)
        ctx.synchronize()
    rec = parity_spot_check(ctx, batch, params, sorted({0, n_tiles // 2, n_tiles - 1}), tile0)
    rec = dict(rank=rank, first_tile=tile0, **rec)
    if distinct and len(chunks) > 1 and rec['result'] == 'bit-exact':
        start = tile0
        for j, c in enumerate(chunks):
            if j:
                batch.synth(SEED, tile0=start)
                batch.classify(params, n_tiles=c)
                ctx.synchronize()
                more = parity_spot_check(ctx, batch, params, sorted({0, c - 1}), start)
                rec['tiles'] += more['tiles']
                if more['result'] != 'bit-exact':
                    rec['result'] = more['result']
                    break
            start += c
        rec['distinct_chunks'] = len(chunks)
    return rec


CHAIN_SUN = dict(azimuth=141.0, elevation=35.0, min_slope_angle=-5.0, max_sun_local_inc_angle=40.0)
CHAIN_FOREST = [111, 113, 115, 116, 121, 123, 125, 126]
CHAIN_MARGIN = 50                  # DEM_MARGIN_IN_PIXELS (:58)
CHAIN_DISTINCT = 4                 # distinct synthetic DEMs / land-cover map pairs, cycled over the tiles


class ChainInputs:
    """Device-resident ancillary rasters of --chain: [n][3760][3760] float32 DEMs, [n][10980][10980] WorldCover and
    [n][3660][3660] CGLS maps (synthetic: proteus_amd.synth), uploaded once before the timed region."""

    def __init__(self, ctx, n_tiles, tile0):
        import numpy as np
        from proteus_amd.synth import synth_dem, synth_landcover_inputs
        self.ctx, self.n, self.tile0 = ctx, n_tiles, tile0
        self.H = self.W = TILE + 2 * CHAIN_MARGIN
        self.host = {}
        for k in range(min(CHAIN_DISTINCT, n_tiles)):
            wc, cg = synth_landcover_inputs(k, TILE, TILE)
            self.host[k] = (synth_dem(k, self.H, self.W), wc, cg)
        dem0, wc0, cg0 = self.host[0]
        self.d_dem, self.d_wc, self.d_cg = (ctx.malloc(n_tiles * a.nbytes) for a in (dem0, wc0, cg0))
        for t in range(n_tiles):
            dem, wc, cg = self.host[self.kind(t)]
            self.d_dem.upload(dem, t * dem.nbytes)
            self.d_wc.upload(wc, t * wc.nbytes)
            self.d_cg.upload(cg, t * cg.nbytes)
        az, zen = np.radians(CHAIN_SUN['azimuth']), np.radians(90.0 - CHAIN_SUN['elevation'])
        self.sun = [np.sin(az) * np.sin(zen), np.cos(az) * np.sin(zen), np.cos(zen)]
        self.sin_az, self.cos_az = np.sin(az), np.cos(az)

    def kind(self, tile):
        return (self.tile0 + tile) % min(CHAIN_DISTINCT, self.n)

    def layers(self, batch):
        """The two layer kernels into the SHAD / LAND planes of `batch` (asynchronous, the context's stream)."""
        self.ctx.shadow_layer_device(self.d_dem.ptr, self.n, self.H, self.W, CHAIN_MARGIN, self.sun, self.sin_az, self.cos_az,
                                     CHAIN_SUN['min_slope_angle'], CHAIN_SUN['max_sun_local_inc_angle'], batch.pin.shad,
                                     float32=True, out_tile_stride=batch.tile_stride)
        self.ctx.landcover_mask_device(self.d_wc.ptr, self.d_cg.ptr, self.n, TILE, TILE, CHAIN_FOREST, batch.pin.land,
                                       out_tile_stride=batch.tile_stride)

    def free(self):
        for b in (self.d_dem, self.d_wc, self.d_cg):
            b.free()


def chain_parity(ctx, batch, params, chain, rank, tile0, tiles):
    """--chain: SHAD against the numpy oracle's terrain shadow layer of the tile's DEM (numpy < 2 promotion, the library's
    default), LAND against the oracle's aggregation of the tile's maps, the seven layers and the counters against the C
    oracle fed with THOSE planes; the reflectance bands, Fmask and the ocean plane against the generator."""
    import numpy as np
    from oracle import c_oracle
    from oracle import dswx_oracle as o
    from proteus_amd import _capi
    from proteus_amd.synth import synth_tile
    cnt = batch.read_counters()
    rec = {'rank': rank, 'first_tile': tile0, 'tiles': [tile0 + t for t in tiles], 'result': 'bit-exact'}
    for tile in tiles:
        dem, wc, cg = chain.host[chain.kind(tile)]
        shad = o.crop_2d_array_all_sides(o.compute_opera_shadow_layer(
            dem, CHAIN_SUN['azimuth'], CHAIN_SUN['elevation'], CHAIN_SUN['min_slope_angle'],
            CHAIN_SUN['max_sun_local_inc_angle'], legacy_promotion=True), CHAIN_MARGIN).astype(np.uint8)
        land = o.landcover_mask_from_warped(wc, cg, CHAIN_FOREST)
        want = synth_tile(tile0 + tile, TILE, TILE, with_masks=True)
        bands = [batch.read_tile(b, tile) for b in _capi.BAND_NAMES]
        fmask, ocean = batch.read_tile('fmask', tile), batch.read_tile('ocean', tile)
        checks = [('SHAD vs the oracle shadow layer', np.array_equal(batch.read_tile('shad', tile), shad)),
                  ('LAND vs the oracle aggregation', np.array_equal(batch.read_tile('land', tile), land)),
                  ('inputs vs the generator', all(np.array_equal(a, b) for a, b in zip(bands, want['bands']))
                   and np.array_equal(fmask, want['fmask']) and np.array_equal(ocean, want['ocean']))]
        exp = c_oracle.classify(params, bands, fmask, land=land, shad=shad, ocean=ocean)
        checks += [(key, np.array_equal(batch.read_tile(key, tile), exp[key]))
                   for key in ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud')]
        checks.append(('counters', cnt[tile].tolist() == exp['counters'].tolist()))
        bad = [name for name, ok in checks if not ok]
        if bad:
            rec['result'] = f'MISMATCH in tile {tile0 + tile}: {bad}'
            break
    return rec


def pmc_traffic(masks, n_tiles):
    """HBM bytes per launch from the committed rocprofv3 PMC passes -- only if they were taken on THIS
    kernel (source hash of dswx_classify_lut.hip + dswx_tables.h + dswx_device.h) and tile count."""
    from proteus_amd import build as _build
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.exists(path):
        return None, 'profiles/pmc_traffic.json absent'
    pmc = json.load(open(path))
    entry = pmc.get('masks' if masks else 'plain')
    if not entry:
        return None, 'no PMC pass for this plane set'
    if entry.get('tiles') != n_tiles:
        return None, f"PMC pass was taken at {entry.get('tiles')} tiles per launch, this run has {n_tiles}"
    now = _build.hot_kernel_hash()
    if entry.get('kernel_source_hash') != now:
        return None, (f"stale: PMC pass taken on kernel sources {entry.get('kernel_source_hash')}, "
                      f'this build is {now} -- re-run tools/run_profiles.sh')
    return entry['hbm_bytes_per_launch'], entry.get('source')


def realloc_spread(ctx, params, n_tiles, masks, repeats, launches=5):
    """The kernel rate depends on where hipMalloc puts the 14 streams: re-allocate
    the batch `repeats` times, time `launches` launches in each, report min / median / max."""
    from proteus_amd import _capi
    from proteus_amd.synth import SEED
    bpp = 24 if masks else 21
    rates = []
    for r in range(repeats):
        b = _capi.DeviceBatch(ctx, n_tiles, TILE, TILE, masks=masks)
        b.synth(SEED)
        b.classify(params)
        ctx.synchronize()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(launches):
            b.classify(params)
        ctx.record(e1)
        ctx.synchronize()
        ms = ctx.elapsed_ms(e0, e1) / launches
        ctx.destroy_event(e0)
        ctx.destroy_event(e1)
        b.free()
        rates.append(n_tiles * TILE * TILE * bpp / (ms * 1e-3) / 1e9)
    rates.sort()
    return {'allocations': repeats, 'launches_each': launches,
            'frac_min': round(rates[0] / HBM_PEAK_GBS, 4),
            'frac_median': round(rates[len(rates) // 2] / HBM_PEAK_GBS, 4),
            'frac_max': round(rates[-1] / HBM_PEAK_GBS, 4)}


def place_batch(ctx, params, n_tiles, tile0, masks, how, trials, refine=1, slack_gib=48.0):
    """Allocate the resident batch through the library (dswx_batch_create) and place it.  The kernel's rate depends on
    WHERE in the address space its output planes lie -- a stable property of the allocation that no layout rule predicts
    from one process to the next (profiles/r03_placement_rule_trials.json) -- so a long-lived batch
    is worth placing, outside the timed region:
      slide   dswx_batch_place_slide: the packed output region timed at offsets 0, 2, ... 48 GiB of a range that much
              longer than itself, four spread layouts, then `refine` passes in which every plane tries the free places
              of the range (HIP virtual memory management: the unused part goes back to the device)
      search  dswx_batch_place_search: one allocation per output plane, each bound to the fastest of `trials` candidates
      first   one allocation per output plane, as they come;   arena   all planes in ONE hipMalloc (a plain caller)"""
    from proteus_amd import _capi
    from proteus_amd.synth import SEED
    rec = {'how': how, 'probes': 0}
    try:
        b = _capi.DeviceBatch(ctx, n_tiles, TILE, TILE, masks=masks, separate_outputs=how in ('search', 'first'),
                              sliding_outputs=how == 'slide')
    except _capi.DswxError as e:
        if how != 'slide' or e.code != _capi.ERR_UNSUPPORTED:
            raise
        # no virtual memory management on this device: the other measured placement
        how, trials = 'search', trials or 6
        rec = {'how': how, 'probes': 0, 'fallback': str(e)[:200]}
        b = _capi.DeviceBatch(ctx, n_tiles, TILE, TILE, masks=masks, separate_outputs=True)
    b.synth(SEED, tile0=tile0)
    ctx.synchronize()
    t_place = time.perf_counter()
    try:
        if how == 'slide':
            rec.update(b.place_slide(params, slack_bytes=int(slack_gib * (1 << 30)), refine_passes=refine))
        elif how == 'search':
            rec.update(b.place_search(params, candidates=trials))
    except Exception as e:          # the placement is an optimisation: the planes bound now are valid whatever happened
        rec['error'] = f'{type(e).__name__}: {e}'[:300]
    rec['seconds'] = round(time.perf_counter() - t_place, 2)
    return b, rec


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as a child
    (torch.distributed.run) and relay its output.  Nothing in this process touches the GPU -- no torch
    import, no HIP call -- so the child ranks are the only GPU users (and nothing here exec()s)."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    return subprocess.run(cmd, env=env).returncode


def rank_plan(case, rank, world):
    """What `rank` of `world` does in one step of `case` (total_tiles, tiles): (tiles it owns, first tile index,
    resident tiles, launches as a list of tile counts).  Shared by the measurement and --plan-only."""
    from proteus_amd import shard
    if case.total_tiles > 0:
        lo, hi = shard.tile_range(case.total_tiles, rank, world)       # this rank's share of the job
        my_tiles = hi - lo
        n_tiles = max(1, min(case.tiles, my_tiles))                    # resident chunk
        return my_tiles, lo, n_tiles, shard.chunk_sizes(my_tiles, n_tiles)
    lo, _ = shard.weak_tile_range(case.tiles, rank)                    # rank r owns tiles [r*T, (r+1)*T)
    return case.tiles, lo, case.tiles, [case.tiles]


def cases_of(args, world):
    """The measured configurations of this command line.  The first is the top-level record of the line.  A plain
    `bench.py --gpus N` (N > 1, neither --tiles nor --total-tiles nor --chain) -- what the driver runs for the scaling
    curve -- measures BOTH: the weak record (256 tiles per GPU = configs[2] x N) and, as the line's `strong` sub-record,
    BASELINE configs[3]: 4096 tiles split over the ranks, walked in 512-tile chunks, every chunk classified once under a
    parity check (--distinct-chunks)."""
    first = argparse.Namespace(total_tiles=args.total_tiles, tiles=args.tiles, distinct_chunks=args.distinct_chunks,
                               key=None)
    cases = [first]
    if (world > 1 and args.plain_command and not args.no_strong) or (args.also_strong and not args.total_tiles and not args.chain):
        cases.append(argparse.Namespace(total_tiles=args.strong_total, tiles=args.strong_chunk, distinct_chunks=True,
                                        key='strong'))
    return cases


def plan_only(args, rank, world):
    from proteus_amd import shard
    cp = shard.ControlPlane(backend='gloo', device=None)
    out = None
    for case in cases_of(args, world):
        my_tiles, tile0, n_tiles, chunks = rank_plan(case, rank, world)
        cp.barrier()
        total = cp.sum_over_ranks(my_tiles)
        plans = cp.gather_objects({'rank': rank, 'first_tile': tile0, 'tiles': my_tiles, 'resident_tiles': n_tiles,
                                   'launches': chunks})
        rec = {'plan_only': True, 'n_gpus': world, 'scaling': 'strong' if case.total_tiles else 'weak',
               'tiles_per_step_all_ranks': total, 'control_plane': cp.backend, 'ranks': plans}
        if case.key is None:
            out = rec
        else:
            out[case.key] = rec
    if rank == 0:
        print(json.dumps(out), flush=True)
    cp.close()
    return 0


def device_identity(torch, index):
    """PCI address and UUID of the device a rank runs on: the line's n_gpus is the number of DISTINCT devices the ranks
    report, not the number of ranks that were started."""
    import socket
    try:
        pr = torch.cuda.get_device_properties(index)
        return f'{socket.gethostname()}/{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}/{pr.uuid}'
    except Exception as e:              # noqa: BLE001
        return f'{socket.gethostname()}/cuda:{index} ({type(e).__name__})'


class RankGuard:
    """one rank's exception must not cost the line.  Every piece of LOCAL work of a rank (allocate,
    place, warm up, the timed loop, the parity check, ...) runs through `run`; the first exception becomes this rank's
    error record and the rest of the rank's local work in this case is skipped -- but the rank keeps walking the same
    sequence of collectives as the others (barriers, MAX, SUM, gathers: they sit OUTSIDE `run`), so nobody is stranded
    and rank 0 still prints the line, with `value: null` for the case a rank failed in and the failure named.
    DSWX_BENCH_INJECT='<rank>:<case index>:<phase>' raises inside that phase (tests/: the injected failure); with ':hard'
    appended the rank's process ENDS there instead (what RankGuard cannot catch: LastWords below)."""

    def __init__(self, rank, case_index, boot_error=None):
        self.rank, self.case_index = rank, case_index
        self.error = dict(boot_error) if boot_error else None
        self.inject, self.inject_hard = None, False
        spec = os.environ.get('DSWX_BENCH_INJECT', '')
        if spec:
            r, c, phase = spec.split(':', 2)
            if int(r) == rank and int(c) == case_index:
                self.inject_hard = phase.endswith(':hard')          # '<rank>:<case>:<phase>:hard' = the process ends there
                self.inject = phase[:-5] if self.inject_hard else phase

    @property
    def ok(self):
        return self.error is None

    def run(self, phase, fn, default=None):
        if self.error is not None:
            return default
        try:
            if self.inject == phase and self.inject_hard:
                os._exit(7)                 # as a GPU fault or a signal would end the rank: no exception, no clean-up
            if self.inject == phase:
                raise RuntimeError(f'injected failure in phase {phase!r} (DSWX_BENCH_INJECT)')
            return fn()
        except Exception as e:                              # noqa: BLE001
            import traceback
            self.error = {'phase': phase, 'error': f'{type(e).__name__}: {e}'[:400]}
            print(f'[bench rank {self.rank}] case {self.case_index}: {phase} failed: {self.error["error"]}\n'
                  + traceback.format_exc(limit=6), file=sys.stderr, flush=True)
            return default


def host_path_leg(ctx, cp, params, check, rank=0, n_tiles=4, reps=3):
    """The END-TO-END (PCIe-inclusive) rate of the host-pointer entry dswx_classify_host, after the timed region and
    never `value`: `n_tiles` 3660 x 3660 tiles (a) from planes in page-locked memory of dswx_host_alloc -- zero copy: the
    kernels read the inputs and write the layers across PCIe themselves -- and (b) from pageable numpy arrays, as a
    caller of the reference's seam holds them (staged copies).  With N ranks every rank runs the same calls AT ONCE
    (barrier on both sides) and the rates are the SUM over ranks: this is north_star's "host scatter / gather" and the one
    place where the GPUs of a node share something (the host's memory system).  A rank whose local work fails keeps
    walking the leg's collectives (RankGuard) and the leg reports the error instead of rates."""
    import ctypes
    import numpy as np
    from proteus_amd import _capi
    from proteus_amd.synth import synth_tile
    layers = ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud')
    guard = RankGuard(rank, 99)
    shape = (n_tiles, TILE, TILE)
    px = n_tiles * TILE * TILE
    rec = {'tiles_per_call': n_tiles, 'ranks': cp.world if cp.dist is not None else 1}
    results = {}
    s = guard.run('host path: generator', lambda: synth_tile(3, TILE, TILE))
    for mode in ('zero_copy', 'pageable'):
        st = {}

        def prepare():
            alloc = ctx.pinned_empty if mode == 'zero_copy' else (lambda sh, dt: np.empty(sh, dtype=dt))
            bands = [alloc(shape, np.int16) for _ in range(6)]
            fm = alloc(shape, np.uint8)
            for t in range(n_tiles):
                for i in range(6):
                    bands[i][t] = s['bands'][i]
                fm[t] = s['fmask']
            outs = {k: alloc(shape, np.uint16 if k == 'diag' else np.uint8) for k in layers}
            for a in outs.values():
                a[...] = 0                          # pageable outputs: touched once, as a caller's arrays would be
            pin, pout = _capi.PlanesIn(), _capi.PlanesOut()
            for i in range(6):
                pin.band[i] = bands[i].ctypes.data
            pin.fmask = fm.ctypes.data
            for k in layers:
                setattr(pout, k, outs[k].ctypes.data)
            st.update(bands=bands, fm=fm, outs=outs, pin=pin, pout=pout, cnt=np.zeros((n_tiles, 3), np.int64))
            call()                                  # untimed: staging arena, tables

        def call():
            _capi._check(ctx.lib.dswx_classify_host(ctx.handle, ctypes.byref(params), n_tiles, TILE, TILE,
                                                    ctypes.byref(st['pin']), ctypes.byref(st['pout']),
                                                    _capi._host_ptr(st['cnt'])))

        guard.run(f'host path: {mode} setup', prepare)
        cp.barrier()
        t0 = time.perf_counter()
        guard.run(f'host path: {mode} calls', lambda: [call() for _ in range(reps)])
        mine = time.perf_counter() - t0
        cp.barrier()
        elapsed = cp.max_over_ranks(mine)
        total_px = cp.sum_over_ranks(px) * reps
        if guard.ok:
            rec[f'{mode}_Gpx_s'] = round(total_px / elapsed / 1e9, 3)
            rec[f'{mode}_ms_per_call'] = round(elapsed / reps * 1e3, 2)
            rec[f'{mode}_kernel'] = ctx.last_kernel_info()
            results[mode] = {k: np.array(v) for k, v in st['outs'].items()}
            results[mode]['counters'] = st['cnt'].copy()
        st.clear()

    def verdict_of():
        same = all(np.array_equal(results['zero_copy'][k], results['pageable'][k]) for k in results['zero_copy'])
        verdict = 'both entries identical' if same else 'MISMATCH between the zero-copy and the staged entry'
        if check and same:
            from oracle import c_oracle
            exp = c_oracle.classify(params, s['bands'], s['fmask'])
            ok = all(np.array_equal(results['zero_copy'][k][n_tiles - 1], exp[k]) for k in layers) and \
                results['zero_copy']['counters'][n_tiles - 1].tolist() == exp['counters'].tolist()
            verdict += ', bit-exact vs the C oracle' if ok else ', MISMATCH vs the C oracle'
        return verdict
    verdict = guard.run('host path: parity', verdict_of)
    mine = {'rank': rank, 'verdict': verdict, 'error': guard.error}
    every = cp.gather_objects(mine)
    failed = [r for r in every if r['error']]
    if failed:          # the rates above are sums over ranks of which one did no work: not rates
        return {'error': f"rank {failed[0]['rank']} failed in {failed[0]['error']['phase']}: {failed[0]['error']['error']}",
                'failed_ranks': [r['rank'] for r in failed], 'tiles_per_call': n_tiles, 'ranks': rec['ranks']}
    rec['pcie_GBps_in_plus_out'] = round(rec['zero_copy_Gpx_s'] * 21, 1)        # 13 B in + 8 B out per pixel, both ways at once
    bad = [r['verdict'] for r in every if 'MISMATCH' in r['verdict']]
    rec['parity'] = bad[0] if bad else verdict
    rec['note'] = ('dswx_classify_host, after the timed region (never `value`): page-locked planes of dswx_host_alloc = zero copy '
                   'across PCIe; pageable numpy planes = staged copies; all ranks at once, rates summed over ranks')
    return rec


def measure_case(args, case, env, case_index=0):
    """One measured configuration on every rank: place the resident batch, warm up, time K steps between barriers,
    check parity on every rank.  Returns (record -- complete on rank 0 --, resident tiles, ranks that failed) and leaves
    nothing allocated.  All local work runs under a RankGuard; the collectives do not (see there)."""
    from proteus_amd import _capi
    ctx, cp, rank, world, params = env.ctx, env.cp, env.rank, env.world, env.params
    guard = RankGuard(rank, case_index, env.boot_error)
    strong = case.total_tiles > 0
    my_tiles, tile0, n_tiles, chunks = rank_plan(case, rank, world)
    how = 'first' if env.share_device and args.placement != 'arena' else args.placement
    st = argparse.Namespace(batch=None, chain=None, placement={'how': how, 'probes': 0}, kernel_info=None,
                            starts=[], stops=[], step_ms=[], chain_split=None)

    def place():
        st.batch, st.placement = place_batch(ctx, params, n_tiles, tile0, args.masks, how, args.placement_trials,
                                             args.slide_refine, env.slack_gib)
    guard.run('place', place)
    if args.chain:
        def chain_inputs():
            st.chain = ChainInputs(ctx, n_tiles, tile0)
        guard.run('chain inputs', chain_inputs)

    def one_step():
        if st.chain:                # terrain shadow + LAND aggregation into the batch's planes, then the classifier
            st.chain.layers(st.batch)
        for c in chunks:            # dswx_batch_classify: a partial last chunk = the first `c` resident tiles
            st.batch.classify(params, n_tiles=c)

    def warm_up():
        for _ in range(args.warmup):
            one_step()
        ctx.synchronize()
        st.kernel_info = ctx.last_kernel_info()
        st.starts = [ctx.event() for _ in range(args.steps)]
        st.stops = [ctx.event() for _ in range(args.steps)]
    guard.run('warm-up', warm_up)

    def timed():
        for k in range(args.steps):
            ctx.record(st.starts[k])
            one_step()
            ctx.record(st.stops[k])
        ctx.synchronize()
        env.device_synchronize()

    cp.barrier()
    guard.run('sync before the timed region', lambda: (env.device_synchronize(), ctx.synchronize()))
    t0 = time.perf_counter()
    guard.run('timed region', timed)
    closing_degraded = cp.barrier()          # True on EVERY rank if an RCCL barrier failed / hung on any (shard.ControlPlane.barrier)
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed

    def after():
        st.step_ms = [ctx.elapsed_ms(a, b) for a, b in zip(st.starts, st.stops)]
        for e in st.starts + st.stops:
            ctx.destroy_event(e)
        st.starts, st.stops = [], []
        if st.chain and rank == 0:          # after the timed region: the three kernels of a step timed one by one
            chain, batch = st.chain, st.batch

            def ms_of(fn, reps=10):
                fn()
                ctx.synchronize()
                e0, e1 = ctx.event(), ctx.event()
                ctx.record(e0)
                for _ in range(reps):
                    fn()
                ctx.record(e1)
                ctx.synchronize()
                ms = ctx.elapsed_ms(e0, e1) / reps
                ctx.destroy_event(e0)
                ctx.destroy_event(e1)
                return ms
            both = ms_of(lambda: chain.layers(batch))
            land_only = ms_of(lambda: ctx.landcover_mask_device(chain.d_wc.ptr, chain.d_cg.ptr, n_tiles, TILE, TILE,
                                                                CHAIN_FOREST, batch.pin.land,
                                                                out_tile_stride=batch.tile_stride))
            st.chain_split = {'terrain_shadow_ms': round(both - land_only, 4), 'land_aggregation_ms': round(land_only, 4),
                              'classify_ms': round(ms_of(lambda: batch.classify(params)), 4)}
    guard.run('read the step timers', after)
    elapsed = cp.max_over_ranks(elapsed)
    total_px_per_step = cp.sum_over_ranks(my_tiles) * TILE * TILE

    bytes_per_px = 24 if args.masks else 21      # 13+8 (16+8 with masks)
    if args.chain:  # + terrain shadow (4 B of DEM incl. its margin + 1 written) + LAND aggregation (9 + 1 + 1 written)
        side = TILE + 2 * CHAIN_MARGIN
        bytes_per_px = 24 + (4.0 * side * side / (TILE * TILE) + 1.0) + 11.0
    # dominant kernel: one launch = one resident chunk; a step is len(chunks) launches
    px_per_launch = n_tiles * TILE * TILE
    my_px_per_step = my_tiles * TILE * TILE
    placement = st.placement

    def probe_frac(ms):
        return round(px_per_launch * bytes_per_px / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms else None

    # what THIS rank saw: gathered into the line, the slowest rank named (value is bounded by it)
    mine = {'rank': rank, 'device': env.device_id, 'tiles_per_step': my_tiles, 'launches_per_step': len(chunks)}
    achieved = avg_launch_ms = None
    if guard.ok:
        avg_step_ms = sum(st.step_ms) / len(st.step_ms)
        achieved = my_px_per_step * bytes_per_px / (avg_step_ms * 1e-3) / 1e9
        avg_launch_ms = avg_step_ms * px_per_launch / my_px_per_step
        mine.update({'wall_ms_per_step': round(my_elapsed / args.steps * 1e3, 4),
                     'launch_ms_avg': round(avg_launch_ms, 4), 'frac': round(achieved / HBM_PEAK_GBS, 4),
                     'placement': {'how': placement['how'], 'probes': placement.get('probes', 0),
                                   'first_come_launch_ms': placement.get('first_come_launch_ms'),
                                   'kept_launch_ms': placement.get('kept_launch_ms'),
                                   'frac_first_come_probe': probe_frac(placement.get('first_come_launch_ms')),
                                   'frac_kept_probe': probe_frac(placement.get('kept_launch_ms'))}})
        for k in ('note', 'error', 'fallback'):
            if placement.get(k):
                mine['placement'][k] = placement[k]
    else:
        mine.update(error=guard.error['error'], phase=guard.error['phase'])
    per_rank = cp.gather_objects(mine)
    failed = [r for r in per_rank if 'error' in r]

    parity = None
    if not args.no_parity:              # EVERY rank checks its own tiles; the records are gathered into the line
        rec = None
        if guard.ok:                    # (the checker failing is reported, not fatal to the measurement)
            try:
                rec = chain_parity(ctx, st.batch, params, st.chain, rank, tile0, sorted({0, n_tiles - 1})) if st.chain else \
                    rank_parity(ctx, st.batch, params, rank, tile0, n_tiles, chunks, case.distinct_chunks)
            except Exception as e:      # noqa: BLE001
                rec = {'rank': rank, 'first_tile': tile0, 'result': f'not checked ({type(e).__name__}: {e})'[:300]}
        else:
            rec = {'rank': rank, 'first_tile': tile0, 'result': f"not checked (the rank failed in {guard.error['phase']})"}
        records = cp.gather_objects(rec)
        bad = [r['result'] for r in records if r['result'] != 'bit-exact']
        parity = {'result': bad[0] if bad else 'bit-exact', 'ranks': records}

    out = None
    if rank == 0:
        from proteus_amd import build as _build
        batch, chain = st.batch, st.chain
        stride = batch.tile_stride if batch is not None else -(-TILE * TILE // 256) * 256
        if strong:
            workload = (f'BASELINE configs[3]: {case.total_tiles} synthetic {TILE}x{TILE} HLS.L30 tiles in all, split '
                        f'contiguously over {world} rank(s); a rank walks its share ({my_tiles} tiles on rank 0) in '
                        f'{len(chunks)} launch(es) over a resident chunk of {n_tiles} tiles (chunks after the first '
                        f're-use the resident planes: same bytes streamed, the generator stays outside the timed region)')
        else:
            workload = (f'BASELINE configs[2]: {n_tiles} synthetic {TILE}x{TILE} HLS.L30 tiles per GPU per step, '
                        f'device-resident band-planar batch')
        if args.chain:
            side = TILE + 2 * CHAIN_MARGIN
            workload = (f"BASELINE configs[4], one GPU's share, device-resident: {n_tiles} synthetic {TILE}x{TILE} tiles per step through "
                        f'terrain shadow layer (DEM {side}x{side}, margin {CHAIN_MARGIN}) -> LAND aggregation (WorldCover '
                        f'{3 * TILE}x{3 * TILE} + CGLS) -> fused classifier with SHAD + LAND + OCEAN, the two layers written straight into '
                        f'the planes of the batch (a step = three kernels; L30 / S30 differ in host-side band mapping only)')
        if args.scaled:
            workload += ' -- flag_offset_and_scale_inputs: the chain on float32 reflectances (0.0001 x), thresholds in those units'
        workload += (f' (tile stride {stride} px = 256-byte aligned tile starts)'
                     + (', LAND+SHAD+OCEAN planes' if args.masks else ''))
        devices = sorted({r['device'] for r in per_rank})
        good = [r for r in per_rank if 'error' not in r]
        # a number that is not a measurement is not printed as one: the barrier that closes the timed region
        # waited for a failing RCCL call (up to its time limit), or several ranks timed one device without saying so
        void = None
        if closing_degraded:
            void = ('the control plane degraded inside the timed region (the closing barrier waited for a failing RCCL call): '
                    + str(cp.backend))[:400]
        elif len(devices) != world and not env.share_device:
            void = (f'{world} ranks ran on {len(devices)} distinct device(s) and DSWX_BENCH_SHARE_DEVICE is not set: '
                    'not a whole-job rate')
        out = {
            'metric': 'Mpixels/sec DSWx classify (3660^2 7-band HLS tiles)',
            # a case in which ANY rank failed has no whole-job rate: null, with the failure named (`error`, `ranks`)
            'value': None if (failed or void) else round(total_px_per_step * args.steps / elapsed / 1e6, 1),
            'unit': 'Mpixels/s',
            # the number of DISTINCT devices the ranks report (PCI address + UUID), not the number of ranks started
            'n_gpus': len(devices), 'n_ranks': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': None if (failed or void) else round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
            'dtype': 'int16+f32' if args.scaled else 'int16+f64', 'data': 'synthetic',
            'rccl_ranks': cp.rccl_ranks,
            'config': {'workload': workload,
                       'tiles_per_step_all_ranks': total_px_per_step // (TILE * TILE),
                       'tiles_per_gpu_resident': n_tiles, 'launches_per_step': len(chunks),
                       'tile': [TILE, TILE], 'tile_stride_px': stride,
                       'planes_in': 10 if args.masks else 7, 'planes_out': 7,
                       'sharding': f'tiles by rank x{world}, no collective',
                       'control_plane': cp.backend,
                       'kernel': st.kernel_info},
            'ranks': per_rank,
            'parity_check': parity,
        }
        if failed:
            out['error'] = '; '.join(f"rank {r['rank']} failed in {r['phase']}: {r['error']}" for r in failed)[:1200]
            out['failed_ranks'] = [r['rank'] for r in failed]
        elif void:
            out['error'] = void
        if good:
            slowest = max(good, key=lambda r: r['wall_ms_per_step'])
            out['slowest_rank'] = {'rank': slowest['rank'], 'device': slowest['device'],
                                   'wall_ms_per_step': slowest['wall_ms_per_step'], 'frac': slowest['frac']}
        if len(devices) != world:
            out['n_gpus_note'] = (f'{world} ranks ran on {len(devices)} distinct device(s)'
                                  + (' (DSWX_BENCH_SHARE_DEVICE=1: a functional test, not a measurement)' if env.share_device else ''))
        if guard.ok:            # rank 0's own kernel figures
            traffic, pmc_note = pmc_traffic(args.masks, n_tiles)
            if chain:
                traffic, pmc_note = None, 'no PMC pass for the three-kernel chain'
            out['config']['arena_placement'] = dict(placement, note={
                'arena': 'all planes in one hipMalloc (dswx_batch_create without flags)',
                'first': 'inputs in one allocation, every output plane in its own, as they come',
                'slide': 'dswx_batch_place_slide (C-ABI): `positions` candidate placements of the output planes inside '
                         'a range 48 GiB longer than they are (packed at every 2 GiB, spread, then per-plane '
                         'refinement), the chunks under the best one moved into a range of their own and kept, the wide range freed; '
                         'first_come_launch_ms = the first-come range timed back to back with the kept one',
                'search': 'dswx_batch_place_search (C-ABI): every output plane in the fastest of `trials` candidate '
                          'allocations (one pass of coordinate descent, the kernel itself as the probe); '
                          'first_come_launch_ms = the first-come planes timed back to back with the kept ones',
            }[placement['how']] + '; before warm-up, outside the timed region; rank 0 (every rank: `ranks`); '
                                  'roofline.realloc_spread shows what unplaced single-arena allocations give'
                + ('; library: ' + placement['note'] if placement.get('note') else ''))
            step_ms = st.step_ms
            out['roofline'] = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                               'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                               'traffic': traffic,
                               'algorithmic_bytes_per_pixel': round(bytes_per_px, 3),
                               'pixels_per_launch': px_per_launch,
                               'launch_ms_avg': round(avg_launch_ms, 4),
                               'launch_ms_min': round(min(step_ms) * px_per_launch / my_px_per_step, 4),
                               'launch_ms_max': round(max(step_ms) * px_per_launch / my_px_per_step, 4),
                               'launches_timed': args.steps * len(chunks),
                               'read_frac_of_peak': round(achieved * (bytes_per_px - 8) / bytes_per_px
                                                          / HBM_PEAK_GBS, 4),
                               'traffic_source': pmc_note,
                               'kernel_source_hash': _build.hot_kernel_hash(),
                               'rank': 0}
            if chain:
                out['roofline'].update(chain=st.chain_split, read_frac_of_peak=None,
                                       note='achieved / frac = the algorithmic bytes of the three kernels of a step (24 + 5.22 + 11 '
                                            'per pixel) over the step time')
            if placement.get('first_come_launch_ms'):
                # the first-come planes and the kept ones, timed back to back at the end of the placement (3-launch probes:
                # compare THESE two with each other; sustained rates over the timed steps run 1.5 - 2 % below such probes)
                out['roofline']['frac_first_come_placement'] = probe_frac(placement['first_come_launch_ms'])
                out['roofline']['frac_kept_placement_probe'] = probe_frac(placement['kept_launch_ms'])
        else:
            out['roofline'] = None
    # nothing stays allocated, whatever happened above
    for obj in (st.chain, st.batch):
        if obj is not None:
            try:
                obj.free()
            except Exception as e:      # noqa: BLE001
                print(f'[bench rank {rank}] free failed: {e}', file=sys.stderr, flush=True)
    return out, n_tiles, [r['rank'] for r in failed]


def resident_bytes(n_tiles, masks):
    """HBM the resident planes of `n_tiles` tiles take (256-byte aligned tile starts; + the counters, negligible)."""
    stride = -(-TILE * TILE // 256) * 256
    return n_tiles * stride * (24 if masks else 21)


def preflight(args, env, cases):
    """VERDICT r04 next-1c, < 10 s, before any big allocation, on every rank: which device (PCI address + UUID), how
    much of its HBM is free against what the largest resident chunk of this command line needs (+ the placement range's
    slack; a rank that is short shrinks ITS slack and says so -- the slack is a per-rank optimisation, no agreement is
    needed), and one control-plane round trip (barrier + MAX).  Gathered; the distinctness of the devices is judged on
    the gathered list.  Returns the record (same on all ranks) and sets env.slack_gib."""
    margin = 6 << 30                                        # probes of the placement keep 8 GiB free themselves
    need = 0
    for case in cases:
        _, _, n_tiles, _ = rank_plan(case, env.rank, env.world)
        need = max(need, resident_bytes(n_tiles, args.masks))
    if args.chain:
        side = TILE + 2 * CHAIN_MARGIN
        need += args.tiles * (4 * side * side + 10 * TILE * TILE)
    mine = {'rank': env.rank, 'device': env.device_id, 'resident_chunk_GiB': round(need / 2 ** 30, 2)}
    env.slack_gib = args.slide_slack_gib
    if env.boot_error:
        mine['error'] = f"{env.boot_error['phase']}: {env.boot_error['error']}"
    else:
        try:
            free, total = env.mem_info()
            mine.update(hbm_free_GiB=round(free / 2 ** 30, 2), hbm_total_GiB=round(total / 2 ** 30, 2))
            slide = args.placement == 'slide' and not env.share_device
            if free < need + margin:
                mine['warning'] = (f'free HBM {free / 2 ** 30:.1f} GiB < resident chunk {need / 2 ** 30:.1f} GiB + '
                                   f'{margin >> 30} GiB: the allocation is likely to fail')
                env.slack_gib = 0.0
            elif slide and free < need + margin + int(args.slide_slack_gib * 2 ** 30):
                env.slack_gib = float(max(0, (free - need - margin) >> 30))
                mine['adjusted'] = f'--slide-slack-gib {args.slide_slack_gib:g} -> {env.slack_gib:g} (free HBM)'
            if slide:
                mine['slide_slack_GiB'] = env.slack_gib
        except Exception as e:                              # noqa: BLE001
            mine['error'] = f'mem info: {type(e).__name__}: {e}'[:300]
    t0 = time.perf_counter()
    env.cp.barrier()
    env.cp.max_over_ranks(0.0)
    mine['control_plane_round_trip_ms'] = round((time.perf_counter() - t0) * 1e3, 3)
    ranks = env.cp.gather_objects(mine)
    devices = [r['device'] for r in ranks]
    rec = {'ranks': ranks, 'distinct_devices': len(set(devices)), 'control_plane': env.cp.backend,
           'rccl_ranks': env.cp.rccl_ranks,
           'ok': len(set(devices)) == len(devices) and not any('error' in r or 'warning' in r for r in ranks)}
    if len(set(devices)) != len(devices):
        rec['note'] = 'two ranks report the same device' + (' (DSWX_BENCH_SHARE_DEVICE=1)' if env.share_device else
                                                             ': the launcher did not give every rank its own GPU')
    return rec


def bring_up(args, rank, local_rank, world):
    """This rank's device, control plane and library context.  The control plane comes first and does not depend on
    the GPU working (gloo underneath, RCCL probed beside it: proteus_amd.shard.ControlPlane), so a rank whose device or
    library fails still takes part in every collective and its failure is a record in the line (`boot_error`)."""
    import torch
    from proteus_amd import build as _build
    boot_error = None
    try:
        # compile when missing or stale (never a CPU fallback, never an old binary); one rank builds
        if local_rank == 0:
            _build.build()                  # os.replace at the end: the file appears complete or not at all
        else:
            for _ in range(1200):
                if not _build.is_stale():
                    break
                time.sleep(0.5)
    except Exception as e:                                  # noqa: BLE001
        boot_error = {'phase': 'build', 'error': f'{type(e).__name__}: {e}'[:400]}
    from proteus_amd import _capi, shard

    share_device = os.environ.get('DSWX_BENCH_SHARE_DEVICE') == '1'
    visible = torch.cuda.device_count()
    if visible < 1 and boot_error is None:
        boot_error = {'phase': 'device', 'error': 'no GPU visible to this rank: the DSWx HIP path has no CPU fallback'}
    if share_device:
        # functional test of the N > 1 code path on a 1-GPU box: every rank on device 0, gloo as
        # the control plane (RCCL refuses two ranks on one GPU).  Not a measurement.
        local_rank = 0
    else:
        # LOCAL_RANK is the device index when every rank sees all GPUs (torchrun's default); a launcher that narrows every
        # rank's view to its own GPU (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES per rank) leaves one device, index 0
        # -- and ONLY then is the index folded (`local_rank % visible` put two ranks on one device without a
        # word when --gpus N exceeded the devices of the box); more ranks than devices otherwise is this rank's failure
        if visible == 1:
            local_rank = 0              # (a 1-GPU box with N > 1 ranks ends here too: preflight names the duplicate
                                        #  devices and the case's value is null unless DSWX_BENCH_SHARE_DEVICE=1)
        elif visible > 1 and local_rank >= visible:
            boot_error = boot_error or {'phase': 'device', 'error': f'LOCAL_RANK {local_rank} but only {visible} GPUs are visible '
                                                                    'to this rank: more ranks than devices'}
            local_rank = local_rank % visible
    device = torch.device('cuda', local_rank)
    if visible >= 1:
        try:
            torch.cuda.set_device(local_rank)
        except Exception as e:                              # noqa: BLE001
            boot_error = boot_error or {'phase': 'device', 'error': f'{type(e).__name__}: {e}'[:400]}
    # N = 1 has no control plane (cp.backend None) unless DSWX_FORCE_DIST=1 asks for a world of one: the RCCL code path
    # of an N > 1 run -- bring-up with device_id, barrier, all_reduce on device tensors, gathers, destroy -- on a box with
    # one GPU (tests/test_gpu_multirank.py)
    cp = shard.ControlPlane(backend='gloo' if share_device else 'nccl', device=None if share_device else device,
                            require=args.require_rccl)
    ctx = None
    if boot_error is None:
        try:
            ctx = _capi.Context(local_rank)        # raises if the HIP extension / GPU is missing
        except Exception as e:                              # noqa: BLE001
            boot_error = {'phase': 'library context', 'error': f'{type(e).__name__}: {e}'[:400]}

    def device_synchronize():
        torch.cuda.synchronize()

    return argparse.Namespace(ctx=ctx, cp=cp, rank=rank, world=world, params=bench_params(args),
                              share_device=share_device, boot_error=boot_error, slack_gib=args.slide_slack_gib,
                              device_id=device_identity(torch, local_rank) if visible >= 1 else f'none (rank {rank})',
                              device_synchronize=device_synchronize,
                              mem_info=lambda: torch.cuda.mem_get_info(local_rank))


class LastWords:
    """Rank 0 of an N > 1 run: what to print if the LAUNCHER ends this process.  An exception on a rank is a record
    (RankGuard); a rank that dies hard -- a signal, a GPU fault that aborts the process -- is not: torchrun then sends
    SIGTERM to the survivors, and rank 0 is usually inside a collective that will never complete (a C call: a Python
    signal handler would not run).  So the C-level handler only writes to a wake-up pipe, and a helper thread that
    blocks on the pipe prints the line as far as it got -- the cases that completed intact, the running one with
    `value: null` and the reason -- and ends the process.  One JSON line on stdout in every ending bench.py can influence."""

    instance = None

    def __init__(self, args, world):
        import signal
        import threading
        LastWords.instance = self
        self.args, self.world = args, world
        self.out = None                     # the line so far (the top-level record once the first case is complete)
        self.running = 'bring-up'
        self.done = False
        self.lock = threading.Lock()
        self.stdout_fd = os.dup(1)          # the REAL stdout: during bring-up fd 1 points at stderr (shard._stdout_to_stderr)
        r, w = os.pipe()
        os.set_blocking(w, False)
        self.pipe = r
        signal.set_wakeup_fd(w, warn_on_full_buffer=False)
        signal.signal(signal.SIGTERM, lambda *a: None)      # (a Python-level handler must exist for the C-level one to run)
        threading.Thread(target=self.watch, daemon=True, name='bench-last-words').start()

    def watch(self):
        import signal
        while True:
            data = os.read(self.pipe, 16)
            if not data:
                return
            if signal.SIGTERM in data:
                self.speak('SIGTERM from the launcher (another rank died?)')

    def speak(self, why):
        with self.lock:
            if self.done:
                return
            self.done = True
            note = f'terminated while {self.running} was running: {why}'
            out = self.out
            if out is None:
                shared = os.environ.get('DSWX_BENCH_SHARE_DEVICE') == '1'
                out = {'metric': 'Mpixels/sec DSWx classify (3660^2 7-band HLS tiles)', 'value': None, 'unit': 'Mpixels/s',
                       'n_gpus': 1 if shared else self.world, 'n_ranks': self.world, 'steps': self.args.steps,
                       'warmup': self.args.warmup, 'ms_per_step': None, 'higher_is_better': True,
                       'scaling': 'strong' if self.args.total_tiles else 'weak', 'vs_baseline': None,
                       'dtype': 'int16+f32' if self.args.scaled else 'int16+f64', 'data': 'synthetic', 'config': {},
                       'roofline': None, 'error': note}
            else:
                out = dict(out)
                if self.running.startswith('case'):
                    out['strong'] = {'value': None, 'error': note}
                else:
                    out['terminated'] = note
            try:
                try:
                    sys.stdout.flush()
                except Exception:           # noqa: BLE001
                    pass
                os.write(self.stdout_fd, (json.dumps(out) + '\n').encode())
            finally:
                os._exit(143)

    def finished(self):
        """The normal ending owns stdout from here on."""
        with self.lock:
            was = self.done
            self.done = True
        return not was


SUB_RECORD_KEYS = ('value', 'unit', 'n_gpus', 'n_ranks', 'steps', 'warmup', 'ms_per_step', 'scaling', 'config', 'roofline',
                   'ranks', 'slowest_rank', 'parity_check', 'error', 'failed_ranks')


def main():
    try:
        return _main()
    except BaseException as e:              # noqa: BLE001
        # what escapes RankGuard is a COLLECTIVE failing (a peer is gone: gloo reports the reset connection at once) or a bug:
        # rank 0 still says what it knows, then the exception takes its course
        lw = LastWords.instance
        if lw is not None and not isinstance(e, SystemExit):
            import traceback
            traceback.print_exc()
            lw.speak(f'{type(e).__name__}: {e}'[:300])
        raise


def _main():
    args = parse_args()
    if args.cpu_parallel_worker:
        cpu_parallel_main(args.cpu_parallel_worker)
        return 0
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # RCCL on this host driver needs dmabuf IPC
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'WORLD_SIZE={world} does not match --gpus {args.gpus}')

    if args.plan_only:
        return plan_only(args, rank, world)

    last_words = LastWords(args, world) if rank == 0 and world > 1 else None
    env = bring_up(args, rank, local_rank, world)
    ctx, cp = env.ctx, env.cp
    cases = cases_of(args, world)
    pre = None
    if world > 1 or args.preflight or cp.dist is not None:
        pre = preflight(args, env, cases)
        if args.preflight:
            cp.barrier()
            if rank == 0:
                print(json.dumps({'preflight': pre}), flush=True)
            if ctx is not None:
                ctx.close()
            cp.close()
            return 0 if pre['ok'] else 1

    out = None
    failed_any = []
    headline_tiles = args.tiles
    for index, case in enumerate(cases):
        if last_words:
            last_words.running = f'case {index}' + (f' ({case.key})' if case.key else '')
        rec, n_tiles, failed = measure_case(args, case, env, index)
        failed_any += failed
        if index + 1 < len(cases) and env.ctx is not None:
            # the next case allocates its own resident chunk (512 tiles = 144 GB for configs[3]): hand the library's pool --
            # the chunks of this case's placed batch and of its dropped ranges, ~100 GiB at 256 tiles -- back to the device
            # first (this process has no other thread that allocates; outside every timed region)
            try:
                from proteus_amd import _capi as _c
                freed = _c.pool_trim()
                if rank == 0:
                    print(f'[bench] pool trimmed between cases: {freed / 2 ** 30:.1f} GiB', file=sys.stderr, flush=True)
            except Exception as e:          # noqa: BLE001
                print(f'[bench rank {rank}] pool trim between cases: {e}', file=sys.stderr, flush=True)
        if rank == 0:
            if case.key is None:
                out = rec
                if pre is not None:
                    out['preflight'] = pre
            else:           # a sub-record: the same fields, minus what only the top level carries
                out[case.key] = {k: rec[k] for k in SUB_RECORD_KEYS if k in rec}
            # what is known so far goes to stderr: a rank that dies HARD in a later case (a signal, not an exception)
            # takes the line with it, this copy stays in the log
            print('[bench partial] ' + json.dumps({k: out.get(k) for k in ('value', 'ms_per_step', 'n_gpus', 'error')}
                                                  | ({case.key: {'value': rec['value']}} if case.key else {})),
                  file=sys.stderr, flush=True)
            if last_words:
                last_words.out = out
                last_words.running = 'the step between two cases'     # a completed record is never overwritten with null
        if case.key is None:
            headline_tiles = n_tiles
    if last_words:
        last_words.running = 'the legs after the timed regions'

    # ---- legs after the timed regions (none of them is `value`)
    host_path = None
    if not args.no_host_path and not args.chain and not failed_any:       # (every rank knows `failed_any`: gathered)
        # every rank takes part (all ranks at once, rates summed); a rank's failure is a record, not an exit
        host_path = host_path_leg(ctx, cp, env.params, check=not args.no_parity, rank=rank)
    # RCCL prints a version banner through C stdio, which is block-buffered when stdout is a pipe and would otherwise
    # come out at process exit, AFTER the JSON line: every rank empties its C buffers now, and rank 0 prints the line
    # behind a barrier, so that it is the last thing on stdout
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:           # noqa: BLE001
        pass
    cp.barrier()
    if rank == 0:
        from proteus_amd import _capi
        out['rccl_ranks'] = cp.rccl_ranks               # as it stands at the END of the run (a late fallback shows)
        out['config']['control_plane'] = cp.backend
        if host_path is not None:
            out['host_path'] = host_path
        solo = world == 1 and not failed_any
        if solo and args.realloc_repeats > 0 and not args.chain:
            try:
                # the legs below allocate whole batches with hipMalloc: hand the library's pool (the chunks of the placed
                # batch's dropped ranges, ~100 GiB at 256 tiles) back to the device first -- this process has no other
                # thread that allocates
                out['roofline']['pool_trimmed_GiB'] = round(_capi.pool_trim() / 2 ** 30, 2)
                out['roofline']['realloc_spread'] = realloc_spread(ctx, env.params, headline_tiles, args.masks,
                                                                   args.realloc_repeats)
            except Exception as e:
                out['roofline']['realloc_spread'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        if solo and not args.no_single_tile and not args.chain:
            try:
                out['single_tile'] = single_tile_leg(ctx, env.params, args.masks)
            except Exception as e:
                out['single_tile'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        if solo and args.plain_command and not args.plain_tiles and not args.no_product_run:
            out['product_run'] = product_run_leg()
        if world == 1 and not args.no_cpu_baseline and not failed_any:
            try:
                out['cpu_baseline'] = cpu_baseline_sample()
                par = cpu_baseline_parallel()
                if par:
                    out['cpu_baseline']['tile_parallel'] = par
            except Exception as e:      # a reported baseline must never cost the bench line
                out['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        if last_words is None or last_words.finished():
            print(json.dumps(out), flush=True)
    code = 1 if failed_any else 0           # non-zero AFTER the line
    try:
        if ctx is not None:
            ctx.close()
        cp.close()
    except Exception as e:                  # noqa: BLE001
        print(f'[bench rank {rank}] shutdown: {e}', file=sys.stderr, flush=True)
    if cp.hung:                             # a helper thread is still inside RCCL: do not wait for it at interpreter exit
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)
    return code


if __name__ == '__main__':
    sys.exit(main())
