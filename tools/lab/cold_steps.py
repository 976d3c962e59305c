#!/usr/bin/env python3
"""Steps of a fresh process's first product run, timed one by one (run it as a fresh process)."""
import json
import os
import sys
import time
t0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
marks = []


def mark(name):
    marks.append((name, round(time.perf_counter() - t0, 4)))


import numpy  # noqa: E402,F401
mark('import numpy')
from proteus_amd import dswx_hls as D, batch, stages  # noqa: E402
mark('import proteus_amd.dswx_hls')
from proteus_amd import _capi, codec  # noqa: E402
lib = _capi.load_library()
mark('load libdswx_hip.so')
ctx = D.get_context(0)
mark('HIP context (dswx_ctx_create)')
codec.load()
mark('load codec')
rc = sys.argv[1]
import logging  # noqa: E402
logging.getLogger('dswx_hls').setLevel(logging.WARNING)
for k in range(3):
    stages.start()
    r = batch._one_tile(D, 0, rc, False)
    st = stages.stop()
    assert r['ok'], r
    mark(f'product {k + 1}')
    if k == 0:
        first = st
print(json.dumps({'marks_s': marks, 'first_product_stages': first}, indent=1))
