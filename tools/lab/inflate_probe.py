#!/usr/bin/env python3
"""Where the inflate time of a band file goes: the same 64 DEFLATE blocks of a coherent-scene band file inflated into
ordinary (pageable) memory and into the page-locked staging buffers the product path uses, on 1 and on all threads.
Prints one JSON object."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_synthetic_hls as synth_hls          # noqa: E402
from proteus_amd import _capi, codec, geotiff   # noqa: E402


def main():
    ctx = _capi.Context(0)
    out = {'cpu_budget': codec.cpu_budget(), 'engine': codec.engine()}
    with tempfile.TemporaryDirectory() as d:
        _, files, _, _ = synth_hls.make(d, scene=True)
        dd = geotiff.open_geotiff(files[0])
        need = dd.n_blocks * dd.block_bytes
        out['file_MB'] = round(len(dd.buf) / 1e6, 2)
        out['raster_MB'] = round(need / 1e6, 2)
        offs = np.asarray(dd.offs[:dd.n_blocks], dtype=np.int64)
        cnts = np.asarray(dd.cnts[:dd.n_blocks], dtype=np.int64)
        for name, buf in (('pageable', np.empty(need, np.uint8)), ('page_locked', ctx.pinned_empty((need,), np.uint8))):
            buf[:] = 0                                   # touch every page first
            for threads in (1, 4, 16):
                best = 1e9
                for _ in range(5):
                    t0 = time.perf_counter()
                    codec.inflate_into(dd.buf, offs, cnts, buf, dd.block_bytes, threads=threads)
                    best = min(best, time.perf_counter() - t0)
                out[f'{name}_threads_{threads}'] = {'ms': round(best * 1e3, 2), 'MBps_out': round(need / best / 1e6, 1)}
        # the copy itself, for scale
        a, b = np.empty(need, np.uint8), ctx.pinned_empty((need,), np.uint8)
        a[:] = 1
        for name, dst in (('memcpy_to_pageable', np.empty(need, np.uint8)), ('memcpy_to_page_locked', b)):
            dst[:] = 0
            t0 = time.perf_counter()
            for _ in range(5):
                np.copyto(dst, a)
            out[name + '_GBps'] = round(5 * need / (time.perf_counter() - t0) / 1e9, 2)
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == '__main__':
    main()
