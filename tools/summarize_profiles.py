#!/usr/bin/env python3
"""Turns the rocprofv3 output of a gpurun call (gpurun_out/prof/...) into the small,
tracked summaries under profiles/:

  profiles/rNN_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (as is)
  profiles/rNN_pmc_counters.csv      per-dispatch FETCH_SIZE / WRITE_SIZE rows of the hot kernel
  profiles/rNN_summary.json          averages + HBM traffic per launch
  profiles/pmc_traffic.json          what bench.py reports as roofline.traffic

HBM bytes follow MI355X_MICROARCH.md section HBM: counters are in KiB; on gfx950
FETCH_SIZE reads exactly half the bytes of a wide coalesced streaming read, so it
is doubled; WRITE_SIZE is exact for streaming stores.  FETCH and WRITE come from
SEPARATE --pmc passes.

usage: tools/summarize_profiles.py ROUND TILES [--masks] [--src gpurun_out/prof]
(the raw rocprofv3 output comes from tools/run_profiles.sh on the GPU box)
"""
import argparse
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)   # newest run wins
    if not hits:
        raise SystemExit(f'no file matches {pattern}')
    return hits[-1]


def counter_rows(src, sub):
    hits = sorted(glob.glob(os.path.join(src, sub, '**', '*counter_collection.csv'), recursive=True),
                  key=os.path.getmtime)
    return list(csv.DictReader(open(hits[-1]))) if hits else []


def short(name):
    return name.split('(')[0].replace('void ', '')


def summarize_next_rows(a, out):
    """Every kernel of tools/next_rows_bench.py (shadow, cover stage 2, land cover, and the fused stage 1):
    kernel-trace average, HBM bytes per launch (FETCH doubled, KiB units, separate passes) and the SQ
    wave-cycle breakdown -> profiles/rNN_next_rows_pmc.json + the kernel-stats csv."""
    tag = f'r{int(a.round):02d}_next_rows'
    stats = one(os.path.join(a.src, 'trace', '**', '*kernel_stats.csv'))
    shutil.copy(stats, os.path.join(out, f'{tag}_kernel_stats.csv'))
    log = os.path.join(a.src, 'next_rows.log')
    if os.path.exists(log):
        txt = open(log).read()
        if '{' in txt:
            open(os.path.join(out, f'{tag}.json'), 'w').write(txt[txt.index('{'):txt.rindex('}') + 1] + '\n')
    trace = list(csv.DictReader(open(one(os.path.join(a.src, 'trace', '**', '*kernel_trace.csv')))))
    res = {}
    for r in trace:
        k = short(r['Kernel_Name'])
        if 'synth' in k or 'build_tables' in k or 'counters_finish' in k:
            continue
        res.setdefault(k, {'durs': []})['durs'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for ctr, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
        for r in counter_rows(a.src, sub):
            k = short(r['Kernel_Name'])
            if k in res and r['Counter_Name'] == ctr:
                res[k].setdefault(ctr, []).append(float(r['Counter_Value']))
    for r in counter_rows(a.src, 'pmc_sq'):
        k = short(r['Kernel_Name'])
        if k in res:
            res[k].setdefault('sq', {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    summary = {}
    for k, v in res.items():
        avg = lambda x: sum(x) / len(x)                      # noqa: E731
        e = {'launches_traced': len(v['durs']), 'avg_ns': avg(v['durs']), 'min_ns': min(v['durs'])}
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            rd, wr = 2.0 * avg(v['FETCH_SIZE']) * 1024.0, avg(v['WRITE_SIZE']) * 1024.0
            e.update(read_bytes_corrected=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr,
                     hbm_GBps_at_trace_avg=(rd + wr) / e['avg_ns'])
        if 'sq' in v:
            sq = {c: avg(x) for c, x in v['sq'].items()}
            wc = sq.get('SQ_WAVE_CYCLES')
            if wc:
                e['sq_fraction_of_wave_cycles'] = {c: round(x / wc, 4) for c, x in sq.items()
                                                   if c not in ('SQ_WAVE_CYCLES', 'SQ_WAVES')}
            e['sq_per_launch_avg'] = sq
        summary[k] = e
    json.dump(summary, open(os.path.join(out, f'{tag}_pmc.json'), 'w'), indent=1)
    print(json.dumps(summary, indent=1))


def summarize_placed(a, out):
    """The kernel trace of `bench.py` in its DEFAULT configuration (--placement slide: dswx_batch_place_slide).
    The trace also holds the placement's ~100 x 4 probe launches -- full-batch launches of the SAME kernel on other plane
    bindings -- and the warm-up, all BEFORE the timed region; after it tools/run_profiles.sh's command line launches
    nothing of the full batch's grid any more (--realloc-repeats 0, --no-single-tile; the parity check launches nothing at
    N = 1 and the host-path leg classifies 4 tiles, a smaller grid).  So the timed region = the LAST `steps` dispatches of
    the fused kernel with the largest grid; the function asserts that they are contiguous in the trace (no other
    full-batch dispatch, and only the counters kernel, between them).  Writes
      profiles/rNN_kernel_stats_placed.csv              rocprofv3-style stats rows over the timed-region dispatches
      profiles/rNN_kernel_stats_placed_all_launches.csv the rocprofv3 --stats summary as it came (probes included)
      profiles/rNN_placed_summary.json                  per-dispatch durations, the bench line of the same run"""
    src = a.src if a.src else os.path.join(ROOT, 'gpurun_out', 'prof_placed')
    tag = f'r{int(a.round):02d}'
    stats = one(os.path.join(src, 'trace', '**', '*kernel_stats.csv'))
    shutil.copy(stats, os.path.join(out, f'{tag}_kernel_stats_placed_all_launches.csv'))
    line = None
    blog = os.path.join(src, 'bench_trace.log')
    for l in open(blog):
        if l.startswith('{"metric"'):
            line = json.loads(l)
    shutil.copy(blog, os.path.join(out, f'{tag}_placed_bench_under_rocprof.log'))
    steps = line['steps'] * line['config']['launches_per_step']
    trace = list(csv.DictReader(open(one(os.path.join(src, 'trace', '**', '*kernel_trace.csv')))))
    trace.sort(key=lambda r: int(r['Start_Timestamp']))
    disp = [r for r in trace if a.kernel in r['Kernel_Name']]
    gmax = max(int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) for r in disp)
    full = [r for r in disp if int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) == gmax]
    timed = full[-steps:]
    t0, t1 = int(timed[0]['Start_Timestamp']), int(timed[-1]['End_Timestamp'])
    # the timed steps run back to back: a gap as long as a launch between two of them would mean that something else
    # (a probe, a re-allocation) sits inside what is taken for the timed region
    gaps = [int(b['Start_Timestamp']) - int(a_['End_Timestamp']) for a_, b in zip(timed, timed[1:])]
    longest = max(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in timed)
    assert all(g < longest for g in gaps), 'the last dispatches of the trace are not one back-to-back timed region'
    rows = {}
    for r in trace:                 # every kernel that ran inside the timed region (the counters kernel too)
        if t0 <= int(r['Start_Timestamp']) and int(r['End_Timestamp']) <= t1:
            rows.setdefault(r['Kernel_Name'], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    total = sum(sum(v) for v in rows.values())
    with open(os.path.join(out, f'{tag}_kernel_stats_placed.csv'), 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), f'{sum(v) / len(v):.1f}', f'{100.0 * sum(v) / total:.4f}', min(v), max(v)])
    durs = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in timed]
    bpp = line['roofline']['algorithmic_bytes_per_pixel']
    px = line['roofline']['pixels_per_launch']
    avg = sum(durs) / len(durs)
    summary = {
        'what': 'rocprofv3 --kernel-trace of `python3 bench.py --tiles 256 --steps 20 --warmup 3` in its default '
                'configuration (--placement slide: dswx_batch_place_slide); the timed region = the last '
                f'{steps} full-batch dispatches of the fused kernel, the earlier ones are the placement probes and the warm-up',
        'kernel': timed[0]['Kernel_Name'], 'full_batch_dispatches_in_trace': len(full), 'timed_region_dispatches': len(durs),
        'trace_avg_ms': avg / 1e6, 'trace_min_ms': min(durs) / 1e6, 'trace_max_ms': max(durs) / 1e6,
        'trace_GBps': px * bpp / avg, 'trace_frac_of_8TBps': px * bpp / avg / 8000.0,
        'bench_line_launch_ms_avg': line['roofline']['launch_ms_avg'], 'bench_line_frac': line['roofline']['frac'],
        'bench_line_ms_per_step': line['ms_per_step'],
        'bench_line_frac_first_come_placement': line['roofline'].get('frac_first_come_placement'),
        'trace_vs_bench_launch_avg': avg / 1e6 / line['roofline']['launch_ms_avg'],
        'placement': line['config']['arena_placement'],
        'timed_dispatch_ns': durs,
    }
    json.dump(summary, open(os.path.join(out, f'{tag}_placed_summary.json'), 'w'), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != 'timed_dispatch_ns'}, indent=1))


def summarize_chain(a, out):
    """Kernel trace of `bench.py --chain`: the timed region = the last `steps` dispatches of each of the step's kernels
    (the placement probes before it launch the classifier only).  Writes profiles/rNN_chain_kernel_stats.csv (stats over the
    timed region) and profiles/rNN_chain_summary.json."""
    tag = f'r{int(a.round):02d}_chain'
    line = None
    blog = os.path.join(a.src, 'bench_trace.log')
    for l in open(blog):
        if l.startswith('{"metric"'):
            line = json.loads(l)
    steps = line['steps']
    trace = list(csv.DictReader(open(one(os.path.join(a.src, 'trace', '**', '*kernel_trace.csv')))))
    trace.sort(key=lambda r: int(r['Start_Timestamp']))
    # the last `steps` shadow-layer dispatches open the timed steps (warm-up comes before, the split timing after
    # the timed region launches each kernel 11 more times: drop those by taking the window of `steps` steps that ends
    # with the classifier dispatch closing the timed region)
    shadow = [r for r in trace if 'dswx_shadow' in r['Kernel_Name']]
    warm = line['warmup']
    timed_shadow = shadow[warm:warm + steps]
    t0 = int(timed_shadow[0]['Start_Timestamp'])
    after = [r for r in trace if int(r['Start_Timestamp']) >= t0]
    rows, counts = {}, {}
    for r in after:
        k = short(r['Kernel_Name'])
        key = 'shadow' if 'shadow' in k else 'land' if 'landcover' in k else 'classify' if 'classify' in k else \
            'counters' if 'counters' in k else None
        if key is None or counts.get(key, 0) >= steps:
            continue
        counts[key] = counts.get(key, 0) + 1
        rows.setdefault(r['Kernel_Name'], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    total = sum(sum(v) for v in rows.values())
    with open(os.path.join(out, f'{tag}_kernel_stats.csv'), 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), f'{sum(v) / len(v):.1f}', f'{100.0 * sum(v) / total:.4f}', min(v), max(v)])
    summary = {'what': 'rocprofv3 --kernel-trace of `python3 bench.py --chain --tiles 256 --steps 20 --warmup 3`: the kernels of the '
                       '20 timed steps', 'step_ms_from_trace': total / steps / 1e6,
               'bench_line_ms_per_step': line['ms_per_step'], 'bench_line_value_Mpix_s': line['value'],
               'bench_line_frac': line['roofline']['frac'], 'bench_line_chain_split_ms': line['roofline']['chain'],
               'kernels_avg_ms': {short(k): sum(v) / len(v) / 1e6 for k, v in rows.items()},
               'parity_check': line['parity_check']['result']}
    json.dump(summary, open(os.path.join(out, f'{tag}_summary.json'), 'w'), indent=1)
    shutil.copy(blog, os.path.join(out, f'{tag}_bench_under_rocprof.log'))
    print(json.dumps(summary, indent=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('round')
    ap.add_argument('tiles', type=int)
    ap.add_argument('--masks', action='store_true')
    ap.add_argument('--next-rows', action='store_true', help='digest gpurun_out/prof_next instead of the hot kernel')
    ap.add_argument('--chain', action='store_true', help='digest gpurun_out/prof_chain: the kernel trace of bench.py --chain')
    ap.add_argument('--placed', action='store_true',
                    help='digest gpurun_out/prof_placed: the kernel trace of the default (placed) bench configuration')
    ap.add_argument('--src', default=None)
    ap.add_argument('--kernel', default='dswx_classify')
    a = ap.parse_args()
    out = os.path.join(ROOT, 'profiles')
    os.makedirs(out, exist_ok=True)
    if a.src is None:
        a.src = os.path.join(ROOT, 'gpurun_out', 'prof_chain' if a.chain else 'prof_placed' if a.placed else 'prof_next' if a.next_rows else ('prof_masks' if a.masks else 'prof'))
    if a.next_rows:
        return summarize_next_rows(a, out)
    if a.placed:
        return summarize_placed(a, out)
    if a.chain:
        return summarize_chain(a, out)
    tag = f'r{int(a.round):02d}' + ('_masks' if a.masks else '')
    stats = one(os.path.join(a.src, 'trace', '**', '*kernel_stats.csv'))
    shutil.copy(stats, os.path.join(out, f'{tag}_kernel_stats.csv'))
    krow = next(r for r in csv.DictReader(open(stats)) if a.kernel in r['Name'])
    # the bench also launches the kernel on ONE tile (single_tile leg): keep only the dispatches of the
    # full batch, i.e. those with the largest grid, taken from the per-dispatch trace
    trace = one(os.path.join(a.src, 'trace', '**', '*kernel_trace.csv'))
    disp = [r for r in csv.DictReader(open(trace)) if a.kernel in r['Kernel_Name']]
    gmax = max(int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) for r in disp)
    durs = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in disp
            if int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) == gmax]
    krow = dict(krow, Calls=len(durs), AverageNs=sum(durs) / len(durs), MinNs=min(durs), MaxNs=max(durs))
    rows = []
    vals = {}
    for ctr, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
        f = one(os.path.join(a.src, sub, '**', '*counter_collection.csv'))
        sel = [r for r in csv.DictReader(open(f))
               if a.kernel in r['Kernel_Name'] and r['Counter_Name'] == ctr]
        big = max(int(r['Grid_Size']) for r in sel)
        sel = [r for r in sel if int(r['Grid_Size']) == big]          # full-batch dispatches only
        rows += sel
        vals[ctr] = [float(r['Counter_Value']) for r in sel]
    keep = ['Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size',
            'VGPR_Count', 'SGPR_Count', 'Counter_Name', 'Counter_Value', 'Start_Timestamp',
            'End_Timestamp']
    with open(os.path.join(out, f'{tag}_pmc_counters.csv'), 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=keep)
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in keep})
    fetch_kib = sum(vals['FETCH_SIZE']) / len(vals['FETCH_SIZE'])
    write_kib = sum(vals['WRITE_SIZE']) / len(vals['WRITE_SIZE'])
    read_bytes = 2.0 * fetch_kib * 1024.0       # gfx950: FETCH_SIZE = half the streamed bytes
    write_bytes = write_kib * 1024.0
    px = a.tiles * 3660 * 3660
    bpp_r, bpp_w = (16 if a.masks else 13), 8
    summary = {
        'round': int(a.round), 'tiles': a.tiles, 'masks': a.masks, 'kernel': krow['Name'],
        'kernel_trace': {'calls': int(krow['Calls']), 'avg_ns': float(krow['AverageNs']),
                         'min_ns': float(krow['MinNs']), 'max_ns': float(krow['MaxNs'])},
        'pmc': {'FETCH_SIZE_KiB_avg': fetch_kib, 'WRITE_SIZE_KiB_avg': write_kib,
                'dispatches': [len(vals['FETCH_SIZE']), len(vals['WRITE_SIZE'])],
                'read_bytes_corrected': read_bytes, 'write_bytes': write_bytes,
                'hbm_bytes_per_launch': read_bytes + write_bytes},
        'algorithmic': {'pixels_per_launch': px, 'read_bytes': px * bpp_r,
                        'write_bytes': px * bpp_w, 'total': px * (bpp_r + bpp_w)},
    }
    summary['traffic_over_algorithmic'] = summary['pmc']['hbm_bytes_per_launch'] / \
        summary['algorithmic']['total']
    summary['achieved_GBps_from_trace'] = summary['algorithmic']['total'] / \
        summary['kernel_trace']['avg_ns']
    # optional third pass: SQ wave-cycle breakdown (quad-cycles; WAIT_ANY + WAIT_INST_ANY +
    # ACTIVE_INST_ANY ~ WAVE_CYCLES, MI355X_MICROARCH.md section rocprofv3 PMC slots)
    sq_files = sorted(glob.glob(os.path.join(a.src, 'pmc_sq', '**', '*counter_collection.csv'), recursive=True),
                      key=os.path.getmtime)
    if sq_files:
        acc = {}
        rows_sq = [r for r in csv.DictReader(open(sq_files[-1])) if a.kernel in r['Kernel_Name']]
        big = max(int(r['Grid_Size']) for r in rows_sq)
        for r in rows_sq:
            if int(r['Grid_Size']) == big:
                acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        sq = {k: sum(v) / len(v) for k, v in acc.items()}
        wc = sq.get('SQ_WAVE_CYCLES')
        summary['sq'] = {'per_launch_avg': sq}
        if wc:
            summary['sq']['fraction_of_wave_cycles'] = {
                k: round(sq[k] / wc, 4) for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY',
                                                  'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS')
                if k in sq}
    json.dump(summary, open(os.path.join(out, f'{tag}_summary.json'), 'w'), indent=1)
    tpath = os.path.join(out, 'pmc_traffic.json')
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    # the hash of the kernel sources the profiled run was built from: bench.py prints it in its line
    khash = None
    blog = os.path.join(a.src, 'bench_trace.log')
    if os.path.exists(blog):
        shutil.copy(blog, os.path.join(out, f'{tag}_bench_under_rocprof.log'))
        for line in open(blog):
            if line.startswith('{"metric"'):
                khash = json.loads(line)['roofline'].get('kernel_source_hash')
    traffic['masks' if a.masks else 'plain'] = {
        'tiles': a.tiles, 'hbm_bytes_per_launch': round(read_bytes + write_bytes),
        'kernel_source_hash': khash,
        'source': f'profiles/{tag}_pmc_counters.csv: rocprofv3 --pmc FETCH_SIZE and --pmc '
                  f'WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950), KiB units'}
    json.dump(traffic, open(tpath, 'w'), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == '__main__':
    main()
