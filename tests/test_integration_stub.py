"""The ctypes stub INTEGRATION.md shows a PROTEUS maintainer is executable documentation: this test
extracts it from the markdown, checks its structure layouts (CPU) and runs it on the GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from proteus_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_source():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r"```python\n(# src/proteus/_dswx_hip\.py\n.*?)```", text, re.S)
    assert m, 'stub block not found in INTEGRATION.md'
    return m.group(1).replace("ctypes.CDLL('libdswx_hip.so')", f"ctypes.CDLL({_capi.library_path()!r})")


def test_stub_structures_match_the_header():
    """Only the declarations (no context is created on a CPU-only box)."""
    src = stub_source()
    decls = src[:src.index('_ctx = ctypes.c_void_p()')]
    ns = {}
    exec(compile(decls, 'INTEGRATION.md', 'exec'), ns)
    for name, ref in (('Params', _capi.Params), ('PlanesIn', _capi.PlanesIn), ('PlanesOut', _capi.PlanesOut)):
        mine = ns[name]
        assert ctypes.sizeof(mine) == ctypes.sizeof(ref), name
        assert [(f[0], getattr(mine, f[0]).offset) for f in mine._fields_] == \
            [(f[0], getattr(ref, f[0]).offset) for f in ref._fields_], name


@pytest.mark.gpu
def test_stub_runs_and_matches_the_oracle():
    from oracle import dswx_oracle as o
    from proteus_amd.synth import synth_tile
    ns = {}
    exec(compile(stub_source(), 'INTEGRATION.md', 'exec'), ns)
    s = synth_tile(77, 150, 210, with_masks=True)
    out, counters = ns['classify'](s['bands'], s['fmask'], ns['default_params'](), land=s['land'],
                                   shad=s['shad'].astype(bool), ocean=s['ocean'])
    exp = o.classify_tile(s['bands'], s['fmask'], landcover=s['land'], shadow=s['shad'], ocean_mask=s['ocean'])
    for layer, key in (('DIAG', 'diag'), ('WTR-1', 'wtr1'), ('WTR-2', 'wtr2'), ('WTR', 'wtr'), ('BWTR', 'bwtr'),
                       ('CONF', 'conf'), ('CLOUD', 'cloud')):
        assert np.array_equal(out[key], exp[layer]), layer
    c = exp['counters']
    assert counters.tolist() == [c['n_valid'], c['n_cloud_and_valid'], c['n_not_ocean']]
    ns['_lib'].dswx_ctx_destroy(ns['_ctx'])
