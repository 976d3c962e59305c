#!/usr/bin/env python3
"""Wall time of `bin/dswx_hls.py runconfig.yaml` as a FRESH process (what the OPERA PGE does per granule): interpreter
start, imports, HIP context, tables, one product.  Prints one JSON object; `-X importtime` of the same command on stderr
of a second run is summarised (top imports)."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_synthetic_hls as synth_hls          # noqa: E402


def main():
    out = {}
    with tempfile.TemporaryDirectory() as d:
        rc = synth_hls.make(d, scene='--noise' not in sys.argv)[0]
        cmd = [sys.executable, os.path.join(ROOT, 'bin', 'dswx_hls.py'), rc]
        runs = []
        for _ in range(4):
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True)
            runs.append(round(time.perf_counter() - t0, 3))
            assert r.returncode == 0, r.stderr[-2000:]
        out['fresh_process_wall_s'] = runs
        t0 = time.perf_counter()
        subprocess.run([sys.executable, '-c', 'pass'])
        out['bare_interpreter_s'] = round(time.perf_counter() - t0, 3)
        r = subprocess.run([sys.executable, '-X', 'importtime'] + cmd[1:], capture_output=True, text=True)
        imports = []
        for line in r.stderr.splitlines():
            if line.startswith('import time:') and '|' in line:
                parts = line.split('|')
                try:
                    imports.append((int(parts[1]), parts[2].strip()))
                except ValueError:
                    pass
        imports.sort(reverse=True)
        out['top_imports_cumulative_us'] = [(n, us) for us, n in imports[:14]]
        log = [ln for ln in r.stderr.splitlines() if 'elapsed' in ln.lower() or 'time' in ln.lower() and 'import time' not in ln]
        out['log_tail'] = log[-3:]
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
