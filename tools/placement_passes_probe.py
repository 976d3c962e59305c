#!/usr/bin/env python3
"""How much do more candidates / a second pass of DeviceBatch.place_outputs buy?  """
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proteus_amd import _capi
from proteus_amd.synth import SEED
ctx = _capi.Context(0)
p = _capi.default_params()
out = []
for cand, passes, launches in ((6, 1, 3), (6, 2, 5), (6, 1, 5), (6, 2, 3), (6, 1, 3), (6, 2, 5)):
    b = _capi.DeviceBatch(ctx, 256, 3660, 3660, separate_outputs=True)
    b.synth(SEED)
    free_bytes, _ = torch.cuda.mem_get_info()
    rec = b.place_outputs(p, candidates=cand, free_bytes=free_bytes, passes=passes, launches=launches)
    # settle: 20 launches
    for _ in range(3):
        b.classify(p)
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(20):
        b.classify(p)
    ctx.record(e1)
    ctx.synchronize()
    ms = ctx.elapsed_ms(e0, e1) / 20
    rec.update(candidates=cand, passes=passes, launches=launches, launch_ms_20=round(ms, 4), frac=round(256 * 3660 * 3660 * 21 / ms / 1e6 / 8000, 4))
    out.append(rec)
    b.free()
print(json.dumps(out, indent=1))
