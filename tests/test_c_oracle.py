"""The scalar C oracle (oracle/dswx_oracle.c) against the reference-generated
goldens, and the exhaustive proof that the HIP kernel's division-free threshold
predicate equals the reference's fl64(n/d) > t (dswx_hls.py:1872-1913)."""
import concurrent.futures as cf

import numpy as np
import pytest

from oracle import c_oracle
from proteus_amd import _capi
from tests import _golden as G

NAME = {'DIAG': 'diag', 'WTR-1': 'wtr1', 'WTR-1-AEROSOL': 'wtr1_aerosol',
        'WTR-2': 'wtr2', 'WTR': 'wtr', 'BWTR': 'bwtr', 'CONF': 'conf', 'CLOUD': 'cloud'}


def params_of_case(c, collapse):
    return _capi.make_params(
        c['thr'], band_fills=c['band_fills'], fmask_fill=c['fmask_fill'],
        mask_adjacent_to_cloud_mode=c['mode'],
        apply_aerosol_class_remapping=c['apply_aerosol'],
        aerosol_fmask_values=c['aerosol_lists'], collapse_wtr_classes=collapse,
        offset_and_scale=c.get('offset_and_scale'))


def check_case(res, c, collapse, name):
    for layer in G.LAYERS:
        key = layer + '.collapsed' if (collapse and layer in G.COLLAPSABLE) else layer
        exp = c['expected'][key]
        got = res[NAME[layer]]
        assert got.dtype == exp.dtype
        assert np.array_equal(got.reshape(exp.shape), exp), (name, layer, collapse)
    assert np.asarray(res['counters']).ravel()[:3].tolist() == \
        c['expected']['counters'][:3].tolist()


@pytest.mark.parametrize('name', [n for n in G.tile_case_names() if 'cover' not in n])
def test_tile_chain_c(name):
    c = G.tile_case(name)
    for collapse in (False, True):
        res = c_oracle.classify(params_of_case(c, collapse), c['bands'], c['fmask'],
                                land=c['land'], shad=c['shad'], ocean=c['ocean'])
        check_case(res, c, collapse, name)


def binary_repr(d):
    return sum(((d >> i) & 1) * 10 ** i for i in range(5)).astype(np.uint16)


@pytest.mark.parametrize('tag', ['default', 'fractional', 'zeros', 'thirds'])
def test_diag_vectors_c(tag):
    z = G.load('diag_vectors.npz')
    cols = [np.ascontiguousarray(z['bands'][:, i]) for i in range(6)]
    p = _capi.make_params(dict(zip(G.THR_KEYS, z['thr_' + tag].tolist())),
                          band_fills=[None] * 6, fmask_fill=None,
                          clip_negative_reflectance=False)
    fm = np.zeros(cols[0].shape, dtype=np.uint8)
    res = c_oracle.classify(p, cols, fm, layers=('diag', 'mndwi', 'ndvi', 'awesh'))
    assert np.array_equal(res['diag'], binary_repr(z['diag_' + tag].ravel()))
    for k in ('mndwi', 'ndvi', 'awesh'):
        assert np.array_equal(res[k], z[k].ravel(), equal_nan=True)


def _predicate_job(args):
    t, lt, lo, hi = args
    return c_oracle.check_quotient_predicate(t, lt, lo, hi)


def test_quotient_predicate_exhaustive():
    """All 2^32 (n, d) int16 pairs, every default threshold and awkward ones."""
    c_oracle.build()
    cases = [(0.124, 0), (-0.44, 0), (-0.5, 0), (0.7, 1),         # defaults
             (0.0, 0), (0.0, 1), (1.0 / 3.0, 0), (2.0 / 3.0, 1), (-1.0, 0), (1.0, 1),
             (0.1, 0), (-0.3, 0), (0.55, 1), (0.25, 0), (0.25, 1), (1e-280, 0), (-1e-280, 1),
             (3.0, 0), (-7.5, 1)]
    chunks = [(-32768 + k * 4096, -32768 + (k + 1) * 4096) for k in range(16)]
    jobs = [(t, lt, lo, hi) for (t, lt) in cases for (lo, hi) in chunks]
    with cf.ProcessPoolExecutor(max_workers=8) as ex:
        results = list(ex.map(_predicate_job, jobs, chunksize=4))
    bad = [(j, r) for j, r in zip(jobs, results) if r[0] != 0]
    assert not bad, bad[:5]


def test_c_oracle_under_ubsan(tmp_path):
    """The scalar C oracle rebuilt with -fsanitize=undefined -fno-sanitize-recover: the golden DIAG
    vectors (int16 extremes, wrap-around sums, n/0 and 0/0 quotients) and a golden tile run clean
    and give the same layers.  In a child process, so that a sanitizer abort is a test failure."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(['make', '-C', os.path.join(root, 'oracle'), 'ubsan'], capture_output=True, text=True)
    if r.returncode != 0 and 'sanitize' in (r.stderr + r.stdout):
        pytest.skip('this gcc has no libubsan')
    assert r.returncode == 0, r.stderr
    code = (
        "import numpy as np\n"
        "from tests import _golden as G\n"
        "from tests.test_c_oracle import params_of_case, check_case\n"
        "from oracle import c_oracle\n"
        "from proteus_amd import _capi\n"
        "z = G.load('diag_vectors.npz')\n"
        "n = z['bands'].shape[0]\n"
        "cols = [np.ascontiguousarray(z['bands'][:, i]).reshape(1, n) for i in range(6)]\n"
        "p = _capi.make_params(dict(zip(G.THR_KEYS, z['thr_fractional'].tolist())), band_fills=[None] * 6,\n"
        "                      fmask_fill=None, clip_negative_reflectance=False)\n"
        "c_oracle.classify(p, cols, np.zeros((1, n), np.uint8))\n"
        "for name in G.tile_case_names()[:3]:\n"
        "    c = G.tile_case(name)\n"
        "    res = c_oracle.classify(params_of_case(c, True), c['bands'], c['fmask'], land=c['land'],\n"
        "                            shad=c['shad'], ocean=c['ocean'])\n"
        "    check_case(res, c, True, name)\n"
        "print('ubsan-clean')\n")
    env = dict(os.environ, DSWX_ORACLE_LIB=os.path.join(root, 'oracle', '_build', 'libdswx_oracle_ubsan.so'),
               PYTHONPATH=root)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0 and 'ubsan-clean' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
