"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/dswx_hip.h declares; parameter packing and error paths that need no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from proteus_amd import _capi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'dswx_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dswx_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build()
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_capi.EXPORTED_SYMBOLS) == names


def test_codec_library_exports_its_header():
    """include/dswx_codec.h (round 6: the DEFLATE side of the GeoTIFF reader / writer, host only): every declared symbol
    is exported by libdswx_codec.so, and nothing else."""
    import subprocess
    text = open(os.path.join(ROOT, 'include', 'dswx_codec.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = sorted(set(re.findall(r'\b(dswx_codec_[a-z0-9_]+)\s*\(', text)))
    assert len(names) == 10
    path = build.build_codec()
    lib = ctypes.CDLL(path)
    for name in names:
        assert hasattr(lib, name), name
    out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True, check=True).stdout
    c_syms = sorted(l.split()[-1] for l in out.splitlines() if ' T ' in l and not l.split()[-1].startswith('_'))
    assert c_syms == names
    lib.dswx_codec_abi_version.restype = ctypes.c_int
    assert lib.dswx_codec_abi_version() == 1
    full = lib.dswx_codec_cpu_budget()
    assert 1 <= full <= (os.cpu_count() or 1)
    lib.dswx_codec_set_cpu_budget(1)
    assert lib.dswx_codec_cpu_budget() == 1
    lib.dswx_codec_set_cpu_budget(10 ** 6)
    assert lib.dswx_codec_cpu_budget() == full          # never more than what was detected
    lib.dswx_codec_set_cpu_budget(0)
    assert lib.dswx_codec_cpu_budget() == full


def test_product_library_carries_no_experiments():
    """the drop-in library exports exactly the C-ABI of include/dswx_hip.h -- no
    probe / lab entry points -- reads no environment switch, and the experiments live in libdswx_lab.so,
    whose own header (tools/lab/csrc/dswx_lab.h) it exports in full."""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', build.build()], capture_output=True, text=True, check=True).stdout
    c_syms = sorted(l.split()[-1] for l in out.splitlines() if ' T ' in l and not l.split()[-1].startswith('_'))
    assert c_syms == declared_symbols()
    assert not any('probe' in l or 'lab_' in l for l in c_syms)
    kernels = [l.split()[-1] for l in out.splitlines() if '_probe_k' in l or 'classify_ws' in l or 'classify_pipe' in l]
    assert kernels == []
    blob = open(build.build(), 'rb').read()
    assert b'DSWX_FUSED_VARIANT' not in blob and b'DSWX_COVER_KERNEL' not in blob and b'getenv' not in blob
    lab = ctypes.CDLL(build.build_lab())
    text = open(os.path.join(ROOT, 'tools', 'lab', 'csrc', 'dswx_lab.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    lab_names = sorted(set(re.findall(r'\b(dswx_[a-z0-9_]+)\s*\(', text)))
    assert lab_names == sorted(_capi.LAB_SYMBOLS)
    for name in lab_names:
        assert hasattr(lab, name), name


def test_abi_version_and_defaults():
    lib = _capi.load_library()
    assert lib.dswx_abi_version() == _capi.DSWX_ABI_VERSION
    p = _capi.default_params()
    # defaults/dswx_hls.yaml:176-212 and :73-101
    assert (p.wigt, p.awgt, p.pswt_1_mndwi, p.pswt_1_ndvi, p.pswt_2_mndwi) == \
        (0.124, 0.0, -0.44, 0.7, -0.5)
    assert (p.pswt_1_nir, p.pswt_1_swir1, p.pswt_2_blue, p.pswt_2_nir, p.pswt_2_swir1,
            p.pswt_2_swir2, p.lcmask_nir) == (1500, 900, 1000, 2500, 3000, 1000, 1200)
    assert list(p.band_fill) == [-9999.0] * 6 and p.fmask_fill == 255.0
    assert p.aerosol_max_nir == 0.1 / 0.0001
    assert p.clip_negative_reflectance == 1 and p.collapse_wtr_classes == 1
    assert p.mask_adjacent_to_cloud_mode == 0 and p.apply_aerosol_class_remapping == 1
    lut = np.array([list(r) for r in p.aerosol_fmask_lut])
    assert sorted(np.nonzero(lut[0])[0]) == [96, 160, 224]
    assert sorted(np.nonzero(lut[1])[0]) == [96, 160, 224]
    assert sorted(np.nonzero(lut[2])[0]) == [96, 128, 160, 192, 224]
    assert sorted(np.nonzero(lut[3])[0]) == [96, 128, 160, 192, 224]


def test_struct_layout_matches_header():
    # sizeof(dswx_params_t): 12+6+1+1 doubles, 10 int32, 4*256 bytes, 6+6 doubles (band_scale, band_offset)
    assert ctypes.sizeof(_capi.Params) == 20 * 8 + 10 * 4 + 1024 + 12 * 8
    assert ctypes.sizeof(_capi.PlanesIn) == 10 * 8
    assert ctypes.sizeof(_capi.PlanesOut) == 12 * 8
    assert ctypes.sizeof(_capi.BatchGeom) == 4 * 8


def test_make_params_field_by_field_against_the_reference_defaults():
    """VERDICT r01 'weak' 3: the C oracle receives its parameters through _capi.make_params, so a wrong
    mapping there would be shared by both sides of every C-oracle comparison.  Here every field of the
    struct make_params builds is checked against values written out by hand from the reference's default
    runconfig (defaults/dswx_hls.yaml:73-101, :176-212) and constants (dswx_hls.py:26, :31, :45-46), for the
    default call and for a call that moves every knob."""
    p = _capi.make_params()
    want = dict(wigt=0.124, awgt=0.0, pswt_1_mndwi=-0.44, pswt_1_nir=1500.0, pswt_1_swir1=900.0, pswt_1_ndvi=0.7,
                pswt_2_mndwi=-0.5, pswt_2_blue=1000.0, pswt_2_nir=2500.0, pswt_2_swir1=3000.0, pswt_2_swir2=1000.0,
                lcmask_nir=1200.0)
    for k, v in want.items():
        assert getattr(p, k) == v, k
    assert list(p.band_fill) == [-9999.0] * 6 and p.fmask_fill == 255.0 and p.aerosol_max_nir == 1000.0
    assert (p.clip_negative_reflectance, p.mask_adjacent_to_cloud_mode, p.apply_aerosol_class_remapping,
            p.collapse_wtr_classes) == (1, 0, 1, 1)
    assert (p.browse_exclude_psw_aggressive, p.browse_not_water_to_nodata, p.browse_cloud_to_nodata,
            p.browse_snow_to_nodata, p.browse_ocean_masked_to_nodata) == (1, 0, 0, 0, 1)
    lut = np.array([list(r) for r in p.aerosol_fmask_lut])
    assert lut.sum() == 3 + 3 + 5 + 5
    assert p.offset_and_scale_inputs == 0 and list(p.band_scale) == [1.0] * 6 and list(p.band_offset) == [0.0] * 6
    r = _capi.make_params(offset_and_scale=[(0.0001, 0.0), (0.0002, 1.0), (1.0, -2.5), (0.5, 0.0), (3.0, 4.0), (1e-4, 7.0)])
    assert r.offset_and_scale_inputs == 1 and list(r.band_scale) == [0.0001, 0.0002, 1.0, 0.5, 3.0, 1e-4]
    assert list(r.band_offset) == [0.0, 1.0, -2.5, 0.0, 4.0, 7.0]
    with pytest.raises(ValueError):
        _capi.make_params(offset_and_scale=[(1.0, 0.0)] * 5)
    # every knob moved
    thr = {k: float(i) + 0.5 for i, k in enumerate(_capi.THRESHOLD_NAMES)}
    q = _capi.make_params(thr, band_fills=[1.0, None, -3.0, 4.5, 0.0, 32767.0], fmask_fill=None,
                          clip_negative_reflectance=False, mask_adjacent_to_cloud_mode='cover',
                          apply_aerosol_class_remapping=False,
                          aerosol_fmask_values={0: [1, 2], 2: [], 3: [255, 0, 300, 7.5], 4: [64]},
                          collapse_wtr_classes=False, aerosol_max_nir=123.25,
                          exclude_psw_aggressive_in_browse=False, not_water_in_browse='nodata',
                          cloud_in_browse='nodata', snow_in_browse='nodata', set_ocean_masked_to_nodata=False)
    for i, k in enumerate(_capi.THRESHOLD_NAMES):
        assert getattr(q, k) == i + 0.5, k
    bf = list(q.band_fill)
    assert bf[0] == 1.0 and np.isnan(bf[1]) and bf[2:] == [-3.0, 4.5, 0.0, 32767.0] and np.isnan(q.fmask_fill)
    assert q.aerosol_max_nir == 123.25
    assert (q.clip_negative_reflectance, q.mask_adjacent_to_cloud_mode, q.apply_aerosol_class_remapping,
            q.collapse_wtr_classes) == (0, 2, 0, 0)
    assert (q.browse_exclude_psw_aggressive, q.browse_not_water_to_nodata, q.browse_cloud_to_nodata,
            q.browse_snow_to_nodata, q.browse_ocean_masked_to_nodata) == (0, 1, 1, 1, 0)
    lut = np.array([list(r) for r in q.aerosol_fmask_lut])
    assert np.nonzero(lut[0])[0].tolist() == [1, 2] and lut[1].sum() == 0
    assert np.nonzero(lut[2])[0].tolist() == [0, 255]           # 300 and 7.5 are not Fmask byte values
    assert np.nonzero(lut[3])[0].tolist() == [64]
    assert _capi.make_params(mask_adjacent_to_cloud_mode='ignore').mask_adjacent_to_cloud_mode == 1
    with pytest.raises(ValueError, match='is not set'):
        _capi.make_params({k: (None if k == 'awgt' else 1.0) for k in _capi.THRESHOLD_NAMES})


_PACKER_C = r"""
/* test-only: fills dswx_params_t BY FIELD NAME from a line-oriented description on stdin and writes the raw
 * struct bytes to stdout -- a packer that shares nothing with proteus_amd/_capi.py but include/dswx_hip.h */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dswx_hip.h"
static double num(void) { char t[64]; if (scanf("%63s", t) != 1) exit(3); return strtod(t, NULL); }
int main(void) {
    char key[64];
    dswx_params_t p;
    memset(&p, 0, sizeof p);
    while (scanf("%63s", key) == 1) {
        if (!strcmp(key, "begin")) memset(&p, 0, sizeof p);
        else if (!strcmp(key, "thr")) {
            p.wigt = num(); p.awgt = num(); p.pswt_1_mndwi = num(); p.pswt_1_nir = num(); p.pswt_1_swir1 = num();
            p.pswt_1_ndvi = num(); p.pswt_2_mndwi = num(); p.pswt_2_blue = num(); p.pswt_2_nir = num();
            p.pswt_2_swir1 = num(); p.pswt_2_swir2 = num(); p.lcmask_nir = num();
        } else if (!strcmp(key, "band_fill")) { for (int i = 0; i < 6; ++i) p.band_fill[i] = num(); }
        else if (!strcmp(key, "fmask_fill")) p.fmask_fill = num();
        else if (!strcmp(key, "aerosol_max_nir")) p.aerosol_max_nir = num();
        else if (!strcmp(key, "clip")) p.clip_negative_reflectance = (int)num();
        else if (!strcmp(key, "mode")) {
            char m[16]; if (scanf("%15s", m) != 1) return 3;
            p.mask_adjacent_to_cloud_mode = !strcmp(m, "mask") ? DSWX_ADJ_MASK : !strcmp(m, "ignore") ? DSWX_ADJ_IGNORE : DSWX_ADJ_COVER;
        } else if (!strcmp(key, "aerosol")) p.apply_aerosol_class_remapping = (int)num();
        else if (!strcmp(key, "collapse")) p.collapse_wtr_classes = (int)num();
        else if (!strcmp(key, "browse")) {
            p.browse_exclude_psw_aggressive = (int)num(); p.browse_not_water_to_nodata = (int)num();
            p.browse_cloud_to_nodata = (int)num(); p.browse_snow_to_nodata = (int)num();
            p.browse_ocean_masked_to_nodata = (int)num();
        } else if (!strcmp(key, "lut")) {             /* row (0..3 = WTR-1 class 0, 2, 3, 4), count, values */
            int row = (int)num(), n = (int)num();
            for (int i = 0; i < n; ++i) p.aerosol_fmask_lut[row][(int)num()] = 1;
        } else if (!strcmp(key, "scale")) {
            p.offset_and_scale_inputs = 1;
            for (int i = 0; i < 6; ++i) { p.band_scale[i] = num(); p.band_offset[i] = num(); }
        } else if (!strcmp(key, "noscale")) {
            for (int i = 0; i < 6; ++i) { p.band_scale[i] = 1.0; p.band_offset[i] = 0.0; }
        } else if (!strcmp(key, "end")) fwrite(&p, sizeof p, 1, stdout);
        else return 2;
    }
    return 0;
}
"""


def test_make_params_against_an_independent_c_packer(tmp_path):
    """_capi.make_params is the one piece of PRODUCT code that both oracles and the GPU path share
    (tests/test_c_oracle.py: params_of_case).  Twenty random runconfig-style parameter sets are packed twice -- by
    make_params, and by a C program compiled here that assigns every field of dswx_params_t by NAME from
    include/dswx_hip.h (the meaning of each runconfig knob restated in this test from dswx_hls.py:2195-2209 fills,
    :1977-1981 modes, :1249-1302 + defaults yaml :77-89 aerosol lists, :5309-5316 browse options, :2295-2302 scale /
    offset) -- and the struct BYTES are compared."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    src = tmp_path / 'packer.c'
    src.write_text(_PACKER_C)
    exe = tmp_path / 'packer'
    subprocess.run(['gcc', '-O1', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    rng = np.random.default_rng(606)
    classes = (0, 2, 3, 4)

    def fmt(v):
        return 'nan' if v != v else repr(float(v))
    text, mine = [], []
    for case in range(20):
        thr = {k: float(rng.choice([rng.normal(0, 1), rng.integers(-3000, 3000), rng.integers(-8, 8) / 8.0]))
               for k in _capi.THRESHOLD_NAMES}
        fills = [None if rng.random() < 0.25 else float(rng.choice([-9999, 0, 32767, -32768, -1000, 1, -9999.5]))
                 for _ in range(6)]
        fmask_fill = None if rng.random() < 0.3 else float(rng.choice([255, 0, 64, 1]))
        mode = str(rng.choice(['mask', 'ignore', 'cover']))
        lists = {c: sorted(set(int(v) for v in rng.integers(0, 256, size=rng.integers(0, 7)))) for c in classes}
        flags = {k: bool(rng.integers(0, 2)) for k in ('clip', 'aerosol', 'collapse', 'psw', 'ocean')}
        browse = {k: str(rng.choice(['nodata', 'white', 'gray'])) for k in ('not_water', 'cloud', 'snow')}
        scale = None if case % 3 else [(float(rng.choice([1e-4, 2e-4, 1.0])), float(rng.integers(-50, 50) / 4.0)) for _ in range(6)]
        max_nir = None if case % 2 else float(rng.integers(1, 4000) / 2.0)
        mine.append(_capi.make_params(
            thr, band_fills=fills, fmask_fill=fmask_fill, clip_negative_reflectance=flags['clip'],
            mask_adjacent_to_cloud_mode=mode, apply_aerosol_class_remapping=flags['aerosol'],
            aerosol_fmask_values=lists, collapse_wtr_classes=flags['collapse'], aerosol_max_nir=max_nir,
            exclude_psw_aggressive_in_browse=flags['psw'], not_water_in_browse=browse['not_water'],
            cloud_in_browse=browse['cloud'], snow_in_browse=browse['snow'], set_ocean_masked_to_nodata=flags['ocean'],
            offset_and_scale=scale))
        # the same set as the packer's input; None (no nodata value: the `image == fill` test can never hold) = NaN
        text += ['begin', 'thr ' + ' '.join(fmt(thr[k]) for k in _capi.THRESHOLD_NAMES),
                 'band_fill ' + ' '.join(fmt(float('nan') if f is None else f) for f in fills),
                 'fmask_fill ' + fmt(float('nan') if fmask_fill is None else fmask_fill),
                 'aerosol_max_nir ' + fmt(0.1 / 0.0001 if max_nir is None else max_nir),      # dswx_hls.py:45-46
                 f"clip {int(flags['clip'])}", f'mode {mode}', f"aerosol {int(flags['aerosol'])}",
                 f"collapse {int(flags['collapse'])}",
                 'browse ' + ' '.join(str(int(v)) for v in (
                     flags['psw'], browse['not_water'] == 'nodata', browse['cloud'] == 'nodata',
                     browse['snow'] == 'nodata', flags['ocean']))]
        for row, c in enumerate(classes):
            text.append(f'lut {row} {len(lists[c])} ' + ' '.join(map(str, lists[c])))
        text.append('noscale' if scale is None else 'scale ' + ' '.join(f'{fmt(a)} {fmt(b)}' for a, b in scale))
        text.append('end')
    out = subprocess.run([str(exe)], input='\n'.join(text).encode(), capture_output=True, check=True).stdout
    size = ctypes.sizeof(_capi.Params)
    assert len(out) == 20 * size
    for i, p in enumerate(mine):
        theirs = out[i * size:(i + 1) * size]
        if bytes(p) != theirs:
            q = _capi.Params.from_buffer_copy(theirs)
            for name, _ in _capi.Params._fields_:
                a, b = getattr(p, name), getattr(q, name)
                a = bytes(a) if hasattr(a, '_length_') else a
                b = bytes(b) if hasattr(b, '_length_') else b
                assert a == b or (a != a and b != b), (i, name, a, b)
            raise AssertionError(f'set {i}: struct bytes differ outside the named fields (padding / NaN payload)')


def test_struct_field_offsets_match_a_c_compiler(tmp_path):
    """The ctypes mirrors in _capi.py against `offsetof` as gcc sees include/dswx_hip.h (sizes alone would not
    notice two swapped fields)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    fields = {'dswx_params_t': [n for n, _ in _capi.Params._fields_],
              'dswx_planes_in_t': [n for n, _ in _capi.PlanesIn._fields_],
              'dswx_planes_out_t': [n for n, _ in _capi.PlanesOut._fields_],
              'dswx_batch_geom_t': [n for n, _ in _capi.BatchGeom._fields_],
              'dswx_batch_layout_t': [n for n, _ in _capi.BatchLayout._fields_],
              'dswx_batch_info_t': [n for n, _ in _capi.BatchInfo._fields_]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dswx_hip.h"', 'int main(void) {']
    for st, names in fields.items():
        for n in names:
            lines.append(f'  printf("{st}.{n} %zu\\n", offsetof({st}, {n}));')
        lines.append(f'  printf("{st}.sizeof %zu\\n", sizeof({st}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'off.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'off'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    mirrors = {'dswx_params_t': _capi.Params, 'dswx_planes_in_t': _capi.PlanesIn,
               'dswx_planes_out_t': _capi.PlanesOut, 'dswx_batch_geom_t': _capi.BatchGeom,
               'dswx_batch_layout_t': _capi.BatchLayout, 'dswx_batch_info_t': _capi.BatchInfo}
    for st, cls in mirrors.items():
        assert int(got[f'{st}.sizeof']) == ctypes.sizeof(cls), st
        for n in fields[st]:
            assert int(got[f'{st}.{n}']) == getattr(cls, n).offset, (st, n)


def test_address_space_budget_account_needs_no_device():
    """dswx_batch_va_budget (ABI v5): the process-wide account of the address space of the sliding ranges -- readable and
    settable without a GPU; nothing is reserved before the first sliding batch."""
    before = _capi.va_budget()
    assert before['budget_bytes'] == 64 << 40 and before['live_bytes'] == 0 and before['retired_bytes'] == 0
    try:
        assert _capi.va_budget(1 << 30)['budget_bytes'] == 1 << 30
        assert _capi.va_budget()['budget_bytes'] == 1 << 30            # 0 = leave as it is
    finally:
        _capi.va_budget(before['budget_bytes'])
    assert _capi.BATCH_ALL_TILES == -1


def test_library_freshness_is_decided_by_content_not_mtime(tmp_path, monkeypatch):
    """a checkout or an rsync reorders mtimes; the stamp beside the library holds a digest of the sources'
    CONTENT and the compiler flags, so a touched source does not make a good prebuilt library stale (a box without
    hipcc would otherwise refuse it) and an edited one does."""
    from proteus_amd import build
    build.build()
    assert not build.is_stale()
    src = build.SOURCES[0]
    st = os.stat(src)
    try:
        os.utime(src, (st.st_atime, st.st_mtime + 10 ** 6))            # "newer" than the library, same bytes
        assert not build.is_stale()
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    edited = tmp_path / 'edited.hip'
    edited.write_bytes(open(src, 'rb').read() + b'\n// edited\n')
    monkeypatch.setattr(build, 'SOURCES', [str(edited)] + build.SOURCES[1:])
    assert build.is_stale()
    monkeypatch.undo()
    monkeypatch.setattr(build, 'HIPCC_FLAGS', build.HIPCC_FLAGS + ['-DSOMETHING'])
    assert build.is_stale()                                            # other flags: another binary


def test_host_entry_points_under_ubsan():
    """SURVEY section 5 (sanitizers): the reference has none; here the C oracle has run under UBSan since round 1, and since
    round 4 the HOST code of the product library does too (proteus_amd.build.build_ubsan: -fsanitize=undefined with
    -fno-sanitize-recover, so the first finding aborts; clang ignores the flag for gfx950 device code).  On a box without
    a GPU that covers the entry points that need none: parameter derivation, the layout rule at the edges of its range,
    the shadow thresholds, the address-space account, the no-device failure.  (The dispatch, batch and placement code runs
    under it on the GPU: tests/test_gpu_parity.py::test_host_code_under_ubsan_on_the_gpu.)"""
    import subprocess
    import sys
    from proteus_amd import build
    code = (
        "import ctypes, itertools, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from proteus_amd import _capi\n"
        "lib = _capi.load_library()\n"
        "assert 'ubsan' in _capi.library_path()\n"
        "p = _capi.default_params()\n"
        "full = {k: getattr(p, k) for k in _capi.THRESHOLD_NAMES}\n"
        "for thr in (None, dict(full, wigt=0.0, awgt=-1e9, pswt_1_nir=1e9, pswt_2_blue=-1e9, lcmask_nir=32767.5)):\n"
        "    _capi.make_params(thr, band_fills=[None, 0.0, 1e6, -32768.0, 32767.0, -9999.5], fmask_fill=300.0,\n"
        "                      aerosol_max_nir=-40000.0, offset_and_scale=[(1e-4, 0.0)] * 6)\n"
        "n = 0\n"
        "for nt, h, w, st, fl in itertools.product((0, 1, 256, 1 << 32, (1 << 32) + 1), (0, 1, 3660, 1 << 30, (1 << 30) + 1),\n"
        "                                          (0, 7, 3660, 1 << 30), (0, 1, 13395600, 13395712, 1 << 46, (1 << 46) + 1),\n"
        "                                          (0, 1, 7, 1 << 10, 1 << 11, 3 << 10, 1 << 20)):\n"
        "    geom = _capi.BatchGeom(nt, h, w, st)\n"
        "    lay = _capi.BatchLayout()\n"
        "    rc = lib.dswx_batch_layout(ctypes.byref(geom), fl, ctypes.byref(lay))\n"
        "    assert rc in (0, -1), rc\n"
        "    n += rc == 0\n"
        "assert n > 100\n"
        "for ms, mi in ((-5, 40), (0, 90), (90, 180), (-91, -1), (1e-300, 1e-300), (89.999999, 179.999999)):\n"
        "    _capi.shadow_thresholds(ms, mi); _capi.shadow_thresholds(ms, mi, float32=True)\n"
        "    t, q = ctypes.c_double(), ctypes.c_double()\n"
        "    lib.dswx_shadow_thresholds(float(ms), float(mi), ctypes.byref(t), ctypes.byref(q))\n"
        "_capi.va_budget(1 << 40); _capi.va_budget((1 << 64) - 1); _capi.va_budget(64 << 40)\n"
        "m = 0\n"      # ABI v6: the COG layout rule at the edges of its range
        "for h, w, eb, tile, fac in itertools.product((0, 1, 29, 3660, (1 << 30) - 1, 1 << 30, 1 << 31), (1, 3660, (1 << 30) - 1, 1 << 40),\n"
        "                                             (0, 1, 2, 3, 4), (0, 8, 16, 100, 512, 4096, 4104), ((), (4, 16, 64, 128), (1, 1, 1), (0,), (1 << 30,), tuple(range(2, 10)))):\n"
        "    lay = _capi.CogLayout()\n"
        "    f = (ctypes.c_int32 * max(len(fac), 1))(*fac)\n"
        "    rc = lib.dswx_cog_layout(h, w, eb, tile, f, len(fac), ctypes.byref(lay))\n"
        "    assert rc in (0, -1), rc\n"
        "    m += rc == 0\n"
        "assert m > 50\n"
        "if _capi.device_count() == 0:\n"
        "    try:\n"
        "        _capi.Context(0)\n"
        "    except _capi.DswxError as e:\n"
        "        assert e.code == _capi.ERR_NO_DEVICE\n"
        "print('UBSAN-CLEAN', n)\n")
    try:
        ubsan_lib = build.build_ubsan()
    except RuntimeError as e:
        pytest.skip(f'no sanitised build on this box: {e}')
    env = dict(os.environ, DSWX_HIP_LIB=ubsan_lib, UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0 and 'UBSAN-CLEAN' in res.stdout, (res.stdout[-500:], res.stderr[-3000:])
    assert 'runtime error' not in res.stderr


def test_bad_mode_raises_like_reference():
    with pytest.raises(Exception, match='ERROR mask adjacent to cloud/cloud-shadow mode'):
        _capi.make_params(mask_adjacent_to_cloud_mode='bogus')


def test_no_device_fails_loudly():
    """On a box without a GPU the context cannot be created -- and nothing falls
    back to a CPU implementation."""
    if _capi.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(_capi.DswxError) as e:
        _capi.Context(0)
    assert e.value.code == _capi.ERR_NO_DEVICE
    assert 'no CPU fallback' in str(e.value)


def test_shadow_thresholds_numpy_and_libm_agree():
    """The pull-back of the two shadow angle tests onto the arccos / arctan arguments
    (dswx_hls.py:4264-4281): numpy's bisection (the Python host) and the library's libm bisection
    (dswx_shadow_thresholds; no GPU needed) find boundaries within a few ulps of each other, and
    each is exact for its own math library."""
    import ctypes
    import numpy as np
    lib = _capi.load_library()
    for ms, mi in [(-5, 40), (0, 90), (10.5, 35.25), (-5, 179.9), (45.0, 1e-3), (-89.999, 120.0)]:
        t_np, q_np = _capi.shadow_thresholds(ms, mi)
        t_c, q_c = ctypes.c_double(), ctypes.c_double()
        assert lib.dswx_shadow_thresholds(float(ms), float(mi), ctypes.byref(t_c), ctypes.byref(q_c)) == 0
        for a, b in ((t_np, t_c.value), (q_np, q_c.value)):
            assert abs(a - b) <= 8 * np.spacing(max(abs(a), abs(b), 1e-300)), (ms, mi, a, b)
        # exactness against numpy on arrays (what the reference evaluates)
        q = np.full(64, q_np)
        assert (np.degrees(np.arccos(q)) <= mi).all()
        assert not (np.degrees(np.arccos(np.nextafter(q, -2.0))) <= mi).any()
        t = np.full(64, t_np)
        assert (np.degrees(np.arctan(t)) <= ms).all()
        assert not (np.degrees(np.arctan(np.nextafter(t, np.inf))) <= ms).any()
    # degenerate thresholds
    assert _capi.shadow_thresholds(-91, -1) == (float('-inf'), 2.0)
    assert _capi.shadow_thresholds(90, 180) == (float('inf'), -1.0)
    t_c, q_c = ctypes.c_double(), ctypes.c_double()
    assert lib.dswx_shadow_thresholds(-91.0, -1.0, ctypes.byref(t_c), ctypes.byref(q_c)) == 0
    assert (t_c.value, q_c.value) == (float('-inf'), 2.0)
    assert lib.dswx_shadow_thresholds(float('nan'), 1.0, ctypes.byref(t_c), ctypes.byref(q_c)) != 0


def test_shadow_thresholds_float32():
    """float32 pull-back (numpy < 2 promotion): exact for numpy's float32 arccos / arctan loops."""
    import numpy as np
    for ms, mi in [(-5, 40), (0, 90), (10.5, 35.25), (-5, 179.9), (45.0, 1e-3)]:
        t, q = _capi.shadow_thresholds(ms, mi, float32=True)
        assert np.float32(t) == t and np.float32(q) == q          # float32 values
        qa = np.full(64, q, np.float32)
        assert (np.degrees(np.arccos(qa)) <= mi).all()
        assert not (np.degrees(np.arccos(np.nextafter(qa, np.float32(-2)))) <= mi).any()
        ta = np.full(64, t, np.float32)
        assert (np.degrees(np.arctan(ta)) <= ms).all()
        assert not (np.degrees(np.arctan(np.nextafter(ta, np.float32(np.inf)))) <= ms).any()
    assert _capi.shadow_thresholds(-91, -1, float32=True) == (float('-inf'), 2.0)
