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


def batch_stub_source():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r"```python\n(# src/proteus/_dswx_hip_batch\.py\n.*?)```", text, re.S)
    assert m, 'batch stub block not found in INTEGRATION.md'
    return m.group(1)


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
    # the resident-batch stub's declarations (everything before its first function)
    bsrc = batch_stub_source()
    ns['ctypes'] = ctypes
    exec(compile(bsrc[:bsrc.index('def _ok(rc):')], 'INTEGRATION.md (batch)', 'exec'), ns)
    for name, ref in (('BatchGeom', _capi.BatchGeom), ('BatchInfo', _capi.BatchInfo)):
        mine = ns[name]
        assert ctypes.sizeof(mine) == ctypes.sizeof(ref), name
        assert [(f[0], getattr(mine, f[0]).offset) for f in mine._fields_] == \
            [(f[0], getattr(ref, f[0]).offset) for f in ref._fields_], name
    assert (ns['DSWX_BATCH_MASKS'], ns['DSWX_BATCH_SEPARATE_OUTPUTS'], ns['DSWX_BATCH_SLIDING_OUTPUTS']) == \
        (_capi.BATCH_MASKS, _capi.BATCH_SEPARATE_OUTPUTS, _capi.BATCH_SLIDING_OUTPUTS)


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


@pytest.mark.gpu
def test_batch_stub_runs_and_matches_the_oracle():
    """the placed, resident batch through the C-ABI alone -- dswx_batch_create,
    dswx_batch_planes, dswx_batch_place_search, dswx_batch_classify, dswx_batch_info, dswx_batch_destroy as the
    INTEGRATION.md stub binds them (no proteus_amd._capi in the call path)."""
    from oracle import c_oracle
    from proteus_amd.synth import SEED, synth_tile
    ns = {}
    exec(compile(stub_source(), 'INTEGRATION.md', 'exec'), ns)
    exec(compile(batch_stub_source(), 'INTEGRATION.md (batch)', 'exec'), ns)
    lib, ctx = ns['_lib'], ns['_ctx']
    n_tiles, h, w = 5, 120, 200
    params = ns['default_params']()

    def upload(geom, pin):
        assert geom.tile_stride == 24064 and geom.tile_stride % 256 == 0          # 24,000 px padded to 256
        assert lib.dswx_synth_batch(ctx, ctypes.c_uint64(SEED), ctypes.c_int64(40), ctypes.byref(geom),
                                    ctypes.byref(pin), None) == 0

    for place in ('slide', 'search', None):
        handle, geom, pin, pout, counters, info = ns['resident_batch'](n_tiles, h, w, params, upload, place=place,
                                                                       slack=24 << 20, step=2 << 20)
        assert info.n_allocations == {'slide': 2, 'search': 8, None: 1}[place] and info.geom.tile_stride == geom.tile_stride
        if place == 'search':
            assert info.search_candidates == 3 and info.search_probes == 2 + 6 * 12
        if place == 'slide':
            assert info.search_probes > 13 + 7                 # 13 packed offsets, the spread layouts, one refinement pass
        if place:
            assert 0 < info.kept_launch_ms <= info.first_come_launch_ms
        else:
            assert info.search_probes == 0
        cnt = np.empty((n_tiles, 3), np.int64)
        assert lib.dswx_memcpy_d2h(ctx, ctypes.c_void_p(cnt.ctypes.data), counters, ctypes.c_size_t(cnt.nbytes)) == 0
        for t in range(n_tiles):
            s = synth_tile(40 + t, h, w)
            exp = c_oracle.classify(params, s['bands'], s['fmask'])
            for key in ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
                got = np.empty((h, w), np.uint16 if key == 'diag' else np.uint8)
                src = getattr(pout, key) + t * geom.tile_stride * got.itemsize
                assert lib.dswx_memcpy_d2h(ctx, ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(src),
                                           ctypes.c_size_t(got.nbytes)) == 0
                assert np.array_equal(got, exp[key]), (key, t, place)
            assert cnt[t].tolist() == exp['counters'].tolist()
        assert lib.dswx_batch_destroy(handle) == 0
    lib.dswx_ctx_destroy(ctx)


def _build_c_example(tmp_path, name='resident_batch'):
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    exe = str(tmp_path / name)
    lib_dir = os.path.dirname(_capi.library_path())
    _capi.load_library()                      # builds the library when missing or stale
    subprocess.run(['gcc', '-std=c11', '-O2', '-Wall', '-Wextra', '-Werror', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'examples', f'{name}.c'), '-L', lib_dir, '-ldswx_hip',
                    f'-Wl,-rpath,{lib_dir}', '-o', exe], check=True)
    return exe


def test_c_example_builds_with_a_c_compiler_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/resident_batch.c: include/dswx_hip.h is a C header (gcc -std=c11 -Wall -Wextra -Werror), the library
    links from C, and without a device the program stops at dswx_ctx_create -- there is no CPU fallback."""
    import subprocess
    exe = _build_c_example(tmp_path)
    if _capi.device_count() > 0:
        pytest.skip('a GPU is present (the GPU test runs the program)')
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and 'no CPU fallback' in r.stderr


@pytest.mark.gpu
def test_c_example_runs_and_matches_the_oracle(tmp_path):
    """The same program on the GPU: counters and FNV-1a checksums of the seven layers of a placed, resident batch,
    produced through the C-ABI from C alone, against the oracle on the same synthetic tiles."""
    import subprocess
    from oracle import c_oracle
    from proteus_amd.synth import synth_tile
    exe = _build_c_example(tmp_path)
    n_tiles, size = 2, 96
    r = subprocess.run([exe, str(n_tiles), str(size)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert f'tile stride {size * size} px' in lines[0] and '2 allocations' in lines[0]       # 9216 = 36 * 256: no padding
    assert lines[1].startswith('address space: ') and ' 0 loose' in lines[1] and 'note' not in lines[1]     # ABI v5 from C
    p = _capi.default_params()
    exp = []
    for t in range(n_tiles):
        s = synth_tile(t, size, size, with_masks=True)
        exp.append(c_oracle.classify(p, s['bands'], s['fmask'], land=s['land'], shad=s['shad'], ocean=s['ocean']))
        c = exp[-1]['counters'].tolist()
        assert f'counters {t}: n_valid {c[0]} n_cloud_and_valid {c[1]} n_not_ocean {c[2]}' in lines

    def fnv(chunks):
        h = 1469598103934665603
        for chunk in chunks:
            for byte in chunk.tobytes():
                h = ((h ^ byte) * 1099511628211) & 0xffffffffffffffff
        return h
    for key in ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
        assert f'checksum {key} {fnv([e[key] for e in exp]):016x}' in lines, key


def test_writer_side_c_example_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import subprocess
    exe = _build_c_example(tmp_path, 'product_layers')
    if _capi.device_count() > 0:
        pytest.skip('a GPU is present (the GPU test runs the program)')
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and 'no CPU fallback' in r.stderr


@pytest.mark.gpu
def test_writer_side_c_example_matches_the_row_by_row_oracle(tmp_path):
    """examples/product_layers.c: the ABI v6 entries (dswx_cog_layout, dswx_cog_blocks_device, dswx_untile_device) from C
    alone -- the level geometry and an FNV-1a checksum of every level's block bytes of the WTR layer against
    oracle/cog_oracle.py (NEAREST pick, tiling, horizontal differencing, row by row) on the oracle's own WTR."""
    import subprocess
    from oracle import c_oracle, cog_oracle
    from proteus_amd.synth import synth_tile
    exe = _build_c_example(tmp_path, 'product_layers')
    size, tile = 700, 128
    r = subprocess.run([exe, str(size), str(tile)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    s = synth_tile(0, size, size, with_masks=True)
    wtr = c_oracle.classify(_capi.default_params(), s['bands'], s['fmask'], land=s['land'], shad=s['shad'], ocean=s['ocean'])['wtr']

    def fnv(data):
        h = 1469598103934665603
        for byte in bytes(data):
            h = ((h ^ byte) * 1099511628211) & 0xffffffffffffffff
        return h
    levels = cog_oracle.cog_levels(wtr, (4, 16, 64, 128), tile, 2)
    assert lines[0] == f'levels {len(levels)}, {sum(len(b) for _, _, b in levels)} bytes'
    for k, ((h, w, data), f) in enumerate(zip(levels, (1, 4, 16, 64, 128))):
        assert lines[1 + k] == (f'level {k}: factor {f}, {h} x {w}, {-(-h // tile)} x {-(-w // tile)} blocks, '
                                f'checksum {fnv(data):016x}'), (k, lines[1 + k])
    assert lines[1 + len(levels)] == f'layer checksum {fnv(wtr.tobytes()):016x}, round trip ok'


def cog_stub_source():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r"```python\n(# src/proteus/_dswx_hip_cog\.py.*?\n.*?)```", text, re.S)
    assert m, 'cog stub block not found in INTEGRATION.md'
    return m.group(1)


def test_cog_stub_structure_matches_the_header():
    ns = {'ctypes': ctypes}
    src = cog_stub_source()
    exec(compile(src[:src.index('def cog_blocks(')], 'INTEGRATION.md (cog)', 'exec'), ns)
    mine, ref = ns['CogLayout'], _capi.CogLayout
    assert ctypes.sizeof(mine) == ctypes.sizeof(ref)
    assert [(f[0], getattr(mine, f[0]).offset) for f in mine._fields_] == [(f[0], getattr(ref, f[0]).offset) for f in ref._fields_]


@pytest.mark.gpu
def test_cog_stub_runs_and_matches_the_host_writer():
    """ABI v6 from the stub of INTEGRATION.md alone: a layer in HBM -> the blocks of the image and of its NEAREST overviews,
    predictor applied == what the host writer (geotiff.blocked_level / overview_nearest) hands to DEFLATE."""
    from proteus_amd import geotiff
    ns = {}
    exec(compile(stub_source(), 'INTEGRATION.md', 'exec'), ns)
    bsrc = batch_stub_source()
    exec(compile(bsrc[bsrc.index('def _ok(rc):'):bsrc.index('def resident_batch(')], 'INTEGRATION.md (batch)', 'exec'), ns)
    exec(compile(cog_stub_source(), 'INTEGRATION.md (cog)', 'exec'), ns)
    lib, ctx = ns['_lib'], ns['_ctx']
    rng = np.random.default_rng(5)
    layer = rng.integers(0, 5, size=(1300, 777)).astype(np.uint8)
    d_layer, d_blocks = ctypes.c_void_p(), ctypes.c_void_p()
    want = [geotiff.blocked_level(lv[None], 512, 2) for lv in [layer] + [geotiff.overview_nearest(layer, f) for f in (4, 16, 64, 128)]]
    total = sum(lv.n_blocks * lv.block_bytes for lv in want)
    assert lib.dswx_device_malloc(ctx, ctypes.c_size_t(layer.nbytes), ctypes.byref(d_layer)) == 0
    assert lib.dswx_device_malloc(ctx, ctypes.c_size_t(total), ctypes.byref(d_blocks)) == 0
    assert lib.dswx_memcpy_h2d(ctx, d_layer, ctypes.c_void_p(layer.ctypes.data), ctypes.c_size_t(layer.nbytes)) == 0
    lay = ns['cog_blocks'](d_layer, 1, 1300, 777, d_blocks)
    assert lib.dswx_stream_synchronize(ctx, None) == 0
    assert lay.n_levels == 5 and lay.total_bytes == total
    got = np.empty(total, np.uint8)
    assert lib.dswx_memcpy_d2h(ctx, ctypes.c_void_p(got.ctypes.data), d_blocks, ctypes.c_size_t(total)) == 0
    for k, lv in enumerate(want):
        n = lv.n_blocks * lv.block_bytes
        assert np.array_equal(got[lay.offset_bytes[k]: lay.offset_bytes[k] + n], lv.data.reshape(-1).view(np.uint8)), k
    lib.dswx_device_free(ctx, d_layer)
    lib.dswx_device_free(ctx, d_blocks)
    lib.dswx_ctx_destroy(ctx)
