#!/usr/bin/env python3
"""Engine / memory clocks and power while the fused kernel runs back to back (rocm-smi polled from a child process)."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from proteus_amd import _capi
from proteus_amd.synth import SEED

samples = []
stop = False


def poll():
    while not stop:
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--json'], capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout)
            card = next(iter(d.values()))
            keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ('sclk', 'mclk', 'fclk', 'power', 'junction', 'hbm', 'memory'))}
            samples.append((round(time.time() - t0, 2), keep))
        except Exception as e:
            samples.append((round(time.time() - t0, 2), {'error': str(e)[:200]}))
        time.sleep(0.3)


lib = [x[4:] for x in sys.argv if x.startswith('LIB=')]
ctx = _capi.Context(0, lib_path=os.path.abspath(lib[0]) if lib else None)      # LIB=path: another build of the product library
p = _capi.default_params()
masks = '--masks' in sys.argv
b = _capi.DeviceBatch(ctx, 256, 3660, 3660, masks=masks)
b.synth(SEED)
ctx.synchronize()
t0 = time.time()
th = threading.Thread(target=poll)
th.start()
time.sleep(1.5)          # idle samples
marks = [('start', round(time.time() - t0, 2))]
times = []
for burst in range(6):
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(60):
        b.classify(p)
    ctx.record(e1)
    ctx.synchronize()
    times.append(round(ctx.elapsed_ms(e0, e1) / 60, 4))
marks.append(('end', round(time.time() - t0, 2)))
time.sleep(1.0)
stop = True
th.join()
short = [[t, s.get('sclk clock speed:'), s.get('mclk clock speed:'), s.get('fclk clock speed:'),
          s.get('Current Socket Graphics Package Power (W)'), s.get('Temperature (Sensor memory) (C)')] for t, s in samples]
print(json.dumps({'launch_ms_per_burst_of_60': times, 'marks': marks,
                  'columns': ['t_s', 'sclk', 'mclk', 'fclk', 'package_power_W', 'hbm_temp_C'], 'samples': short}))
