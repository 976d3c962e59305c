#!/usr/bin/env python3
"""What a long-lived service does to the device: `--cycles` times create a sliding batch of
`--tiles` 3660 x 3660 tiles, generate, place (dswx_batch_place_slide), classify, check one tile against the C oracle,
destroy -- and after every cycle record the device's free memory and the library's address-space account.  Memory must
stay bounded (the chunks of dropped ranges go into the library's pool and are what the next cycle's ranges are built from;
dswx_batch_pool_trim at the end gives the pool back); address space is consumed at 100 - 160 GiB per cycle at 256 tiles
until the budget is reached, after which batches are allocated packed (the note says
so) and the service keeps running.  `--budget-gib` caps the budget so that the fallback is reached inside the run.

    python tests/helpers/service_cycles_soak.py --tiles 256 --cycles 12 --budget-gib 800

(Lives under tests/ because it uses the oracle as its checker; its name keeps pytest from collecting it.)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle                         # noqa: E402  (checker)
from proteus_amd import _capi                       # noqa: E402
from proteus_amd.synth import SEED, synth_tile      # noqa: E402


def free_bytes():
    hip = ctypes.CDLL('libamdhip64.so')
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tiles', type=int, default=256)
    ap.add_argument('--cycles', type=int, default=12)
    ap.add_argument('--budget-gib', type=float, default=0.0)
    a = ap.parse_args()
    ctx = _capi.Context(0)
    p = _capi.default_params()
    if a.budget_gib:
        _capi.va_budget(int(a.budget_gib * 2 ** 30))
    T = 3660
    px = a.tiles * T * T
    f_start = free_bytes()
    rows = []
    for c in range(a.cycles):
        t0 = time.perf_counter()
        b = _capi.DeviceBatch(ctx, a.tiles, T, T, sliding_outputs=True)
        b.synth(SEED, tile0=c)
        rec = b.place_slide(p)
        info = b.info()
        e0, e1 = ctx.event(), ctx.event()
        b.classify(p)
        ctx.synchronize()
        ctx.record(e0)
        for _ in range(5):
            b.classify(p)
        ctx.record(e1)
        ctx.synchronize()
        ms = ctx.elapsed_ms(e0, e1) / 5
        ctx.destroy_event(e0)
        ctx.destroy_event(e1)
        t = a.tiles - 1
        s = synth_tile(c + t, T, T)
        exp = c_oracle.classify(p, s['bands'], s['fmask'])
        exact = all(np.array_equal(b.read_tile(k, t), exp[k]) for k in ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud')) \
            and b.read_counters()[t].tolist() == exp['counters'].tolist()
        held = f_start - free_bytes()
        b.free()
        acct = _capi.va_budget()
        rows.append({'cycle': c, 'sliding': bool(info['flags'] & _capi.BATCH_SLIDING_OUTPUTS), 'positions': rec['positions'],
                     'frac_of_8TBps': round(px * 21 / (ms * 1e-3) / 8e12, 4), 'bit_exact': bool(exact),
                     'held_with_batch_GiB': round(held / 2 ** 30, 2), 'batch_GiB': round(info['bytes_allocated'] / 2 ** 30, 2),
                     'free_lost_after_destroy_MiB': round((f_start - free_bytes()) / 2 ** 20, 1),
                     'pooled_GiB': round(acct['pooled_bytes'] / 2 ** 30, 2),
                     'va_quarantined_GiB': round(acct['retired_bytes'] / 2 ** 30, 1), 'va_loose_GiB': round(acct['loose_bytes'] / 2 ** 30, 1),
                     'note': info['note'][:120], 'seconds': round(time.perf_counter() - t0, 1)})
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    # memory not back on the device after a destroy = the pool (reused by the next cycle) + the context's own scratch; it
    # must stop growing once the pool holds what one cycle needs
    unaccounted = [r['free_lost_after_destroy_MiB'] - r['pooled_GiB'] * 1024 for r in rows]
    released = _capi.pool_trim()
    after_trim = (f_start - free_bytes()) / 2 ** 20
    ok = all(r['bit_exact'] for r in rows) and max(unaccounted) - min(unaccounted) < 64 and after_trim < 128 and \
        max(r['free_lost_after_destroy_MiB'] for r in rows[1:]) <= rows[1]['free_lost_after_destroy_MiB'] + 4096
    print(json.dumps({'ok': bool(ok), 'tiles': a.tiles, 'cycles': a.cycles, 'budget_GiB': a.budget_gib or 65536,
                      'free_at_start_GiB': round(f_start / 2 ** 30, 2), 'pool_trim_released_GiB': round(released / 2 ** 30, 2),
                      'free_lost_after_trim_MiB': round(after_trim, 1), 'account_after_trim': _capi.va_budget(), 'rows': rows}))
    ctx.close()
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
