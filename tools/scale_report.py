#!/usr/bin/env python3
"""Reads bench.py lines of several GPU counts (files, or a SCALE_rNN.json-like list) and prints what north_star asks for
once a multi-GPU node has run them: Mpixels/s at 1 / 2 / 4 / 8 GPUs as absolute numbers and as achieved fraction of the
HBM roofline (per GPU, from each rank's own launch times), weak-scaling efficiency against N = 1, the slowest rank, the
control plane that carried the barriers, and whatever failed.

    python tools/scale_report.py line_n1.json line_n2.json line_n4.json line_n8.json
    python tools/scale_report.py SCALE_r05.json            # a JSON list / dict of lines, or of {"parsed": line} records

No GPU, no oracle: it only reads the lines."""
import json
import sys


def lines_of(path):
    txt = open(path).read().strip()
    try:
        obj = json.loads(txt)
    except ValueError:                      # a log: keep the bench lines
        return [json.loads(x) for x in txt.splitlines() if x.startswith('{"metric"')]
    found = []

    def walk(o):
        if isinstance(o, dict):
            if 'metric' in o and 'n_gpus' in o:
                found.append(o)
            else:
                for v in o.values():
                    walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
    walk(obj)
    return found


def report(lines):
    lines = sorted(lines, key=lambda d: d.get('n_ranks', d.get('n_gpus', 1)))
    base = next((d for d in lines if d.get('n_ranks', d.get('n_gpus')) == 1 and d.get('value')), None)
    rows = []
    for d in lines:
        n = d.get('n_ranks', d.get('n_gpus', 1))
        ranks = [r for r in d.get('ranks', []) if 'frac' in r]
        fracs = [r['frac'] for r in ranks] or ([d['roofline']['frac']] if d.get('roofline') else [])
        row = {'ranks': n, 'distinct_gpus': d.get('n_gpus'), 'value_Mpx_s': d.get('value'),
               'frac_of_HBM_peak_per_gpu_min_mean_max': [round(min(fracs), 4), round(sum(fracs) / len(fracs), 4),
                                                         round(max(fracs), 4)] if fracs else None,
               'efficiency_vs_n1': round(d['value'] / (n * base['value']), 4) if base and d.get('value') else None,
               'speedup_vs_n1': round(d['value'] / base['value'], 3) if base and d.get('value') else None,
               'slowest_rank': (d.get('slowest_rank') or {}).get('rank'),
               'control_plane': (d.get('config') or {}).get('control_plane'), 'rccl_ranks': d.get('rccl_ranks'),
               'parity': (d.get('parity_check') or {}).get('result'), 'error': d.get('error')}
        st = d.get('strong')
        if st:
            row['strong_4096_tiles'] = {'value_Mpx_s': st.get('value'), 'ms_per_step': st.get('ms_per_step'),
                                        'parity': (st.get('parity_check') or {}).get('result'), 'error': st.get('error')}
        if d.get('n_gpus') != n:
            row['note'] = d.get('n_gpus_note', f'{n} ranks on {d.get("n_gpus")} distinct device(s)')
        rows.append(row)
    return {'metric': lines[0].get('metric') if lines else None, 'baseline_n1_Mpx_s': base['value'] if base else None,
            'rows': rows}


def main():
    lines = []
    for path in sys.argv[1:]:
        lines += lines_of(path)
    if not lines:
        raise SystemExit('no bench line found')
    print(json.dumps(report(lines), indent=1))


if __name__ == '__main__':
    main()
