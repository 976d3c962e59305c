#!/usr/bin/env python3
"""is dswx_batch_pool_trim safe while a PLACED batch is live?  After a kept dswx_batch_place_slide the
batch's chunks were moved (rehome) out of the wide range, which is then retired; trim frees the wide range's RESERVATION
while chunks that were once mapped in it are mapped in the live batch's range.  This probe does exactly that, `rounds`
times: sliding batch, place (repeat until a placement is kept), trim while live, re-generate the inputs with other tile
indices, classify, compare every layer of every tile with the C oracle, and report hipMemGetInfo around the trim.
Prints one JSON object."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                # noqa: E402
from oracle import c_oracle                       # noqa: E402
from proteus_amd import _capi                     # noqa: E402
from proteus_amd.synth import SEED, synth_tile    # noqa: E402


def free_bytes():
    hip = ctypes.CDLL('libamdhip64.so')
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n, h, w = 6, 1024, 1024
    ctx = _capi.Context(0)
    p = _capi.default_params()
    out = {'rounds': []}
    for r in range(rounds):
        b = _capi.DeviceBatch(ctx, n, h, w, sliding_outputs=True)
        b.synth(SEED, tile0=100 * r)
        kept = False
        tries = 0
        while not kept and tries < 8:
            before = b.info()['va_reserved_bytes']
            rec = b.place_slide(p, slack_bytes=256 << 20, step_bytes=16 << 20, launches=2)
            kept = b.info()['va_reserved_bytes'] != before or rec['kept_launch_ms'] < rec['first_come_launch_ms']
            tries += 1
        f0 = free_bytes()
        acct0 = _capi.va_budget()
        released = _capi.pool_trim()                # <-- while the placed batch is live
        f1 = free_bytes()
        b.synth(SEED, tile0=100 * r + 50)           # new inputs through the same planes
        b.classify(p)
        ctx.synchronize()
        cnt = b.read_counters()
        bad = []
        for t in range(n):
            s = synth_tile(100 * r + 50 + t, h, w)
            e = c_oracle.classify(p, s['bands'], s['fmask'])
            for k in ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
                if not np.array_equal(b.read_tile(k, t), e[k]):
                    bad.append([t, k])
            if cnt[t].tolist() != e['counters'].tolist():
                bad.append([t, 'counters'])
        b.free()
        released2 = _capi.pool_trim()
        f2 = free_bytes()
        out['rounds'].append({'kept_after_tries': tries if kept else None, 'released_live_MiB': released >> 20,
                              'free_before_trim_MiB': f0 >> 20, 'free_after_live_trim_MiB': f1 >> 20,
                              'free_after_batch_freed_and_trim_MiB': f2 >> 20, 'released_after_MiB': released2 >> 20,
                              'account_before': acct0, 'account_after': _capi.va_budget(), 'mismatches': bad})
    out['wrong_rounds'] = sum(1 for x in out['rounds'] if x['mismatches'])
    print(json.dumps(out))
    ctx.close()


if __name__ == '__main__':
    main()
