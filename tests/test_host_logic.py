"""Host-side logic that needs no GPU: GeoTIFF I/O, runconfig handling, CLI, metadata,
product comparison and the error conventions of the reference interface."""
import ctypes
import os
import struct
import zlib
import sys

import numpy as np
import pytest
import yaml

from proteus_amd import dswx_hls as D
from proteus_amd import geotiff, runconfig as rc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_synthetic_hls as synth_hls   # noqa: E402


# ---- GeoTIFF -------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [np.uint8, np.int16, np.uint16, np.float32, np.float64])
@pytest.mark.parametrize('compress', [True, False])
def test_geotiff_roundtrip(tmp_path, dtype, compress):
    rng = np.random.default_rng(1)
    a = (rng.integers(0, 200, size=(700, 531)) - 50).astype(dtype)
    if np.issubdtype(dtype, np.floating):
        a = a + dtype(0.25)
        a[3, 4] = np.nan
    geo = geotiff.geo_tags_from_geotransform((5e5, 30.0, 0, 4e6, 0, -30.0), epsg=32615)
    path = str(tmp_path / 'x.tif')
    nod = float('nan') if np.issubdtype(dtype, np.floating) else 255
    geotiff.write_geotiff(path, a, geo_tags=geo, metadata={'A': 'b & <c>', 'N': 5}, nodata=nod,
                          descriptions=['Layer "one"'], compress=compress)
    b, info = geotiff.read_geotiff(path)
    assert b.dtype == a.dtype and np.array_equal(a, b, equal_nan=True)
    assert info.geotransform == (5e5, 30.0, 0.0, 4e6, 0.0, -30.0)
    assert info.metadata == {'A': 'b & <c>', 'N': '5'}
    assert info.descriptions == ['Layer "one"']
    assert (np.isnan(info.nodata) if np.issubdtype(dtype, np.floating) else info.nodata == 255.0)
    assert info.geo_tags[geotiff.TAG_GEOKEYS][1][-1] == 32615
    sub, _ = geotiff.read_geotiff(path, window=(0, 0, 100, 50))
    assert np.array_equal(sub, a[:50, :100], equal_nan=True)


def test_geotiff_multiband_and_colormap(tmp_path):
    rng = np.random.default_rng(2)
    stack = rng.integers(0, 256, size=(3, 100, 90)).astype(np.uint8)
    p = str(tmp_path / 'm.tif')
    geotiff.write_geotiff(p, stack, descriptions=['a', 'b', 'c'], nodata=255)
    back, info = geotiff.read_geotiff(p)
    assert info.bands == 3 and np.array_equal(back, stack) and info.descriptions == ['a', 'b', 'c']
    p2 = str(tmp_path / 'c.tif')
    geotiff.write_geotiff(p2, stack[0], colormap={0: (255, 255, 255), 1: (0, 0, 255), 255: (0, 0, 0)})
    _, info = geotiff.read_geotiff(p2)
    assert info.colormap[1].tolist() == [0, 0, 255] and info.colormap[0].tolist() == [255, 255, 255]


@pytest.mark.parametrize('shape', [(3660, 3660), (700, 531), (513, 40)])
def test_cog_overviews_and_layout(tmp_path, shape):
    """save_as_cog (reference core.py:7-91): NEAREST overviews 4/16/64/128, all IFDs first, block
    data from the smallest overview to the full-resolution image; checked with the rules of the
    reference's validator (extern/validate_cloud_optimized_geotiff.py:176-297)."""
    rng = np.random.default_rng(3)
    a = rng.integers(0, 6, size=shape).astype(np.uint8)
    p = str(tmp_path / 'cog.tif')
    geotiff.write_geotiff(p, a, nodata=255, colormap={0: (255, 255, 255), 1: (0, 0, 255)},
                          metadata={'K': 'v'}, overviews=geotiff.COG_OVERVIEW_FACTORS)
    assert geotiff.validate_cog(p) == []
    lv = geotiff.cog_layout(p)
    assert [(x['height'], x['width']) for x in lv] == \
        [shape] + [(-(-shape[0] // f), -(-shape[1] // f)) for f in geotiff.COG_OVERVIEW_FACTORS]
    assert lv[0]['ifd_offset'] == 8 and all(x['tile'] == (512, 512) for x in lv)
    back, info = geotiff.read_geotiff(p)
    assert np.array_equal(back, a) and info.metadata == {'K': 'v'} and info.nodata == 255.0
    for k, f in enumerate(geotiff.COG_OVERVIEW_FACTORS):
        o, oi = geotiff.read_geotiff(p, overview=k)
        h, w = o.shape
        ys = (0.5 + np.arange(h) * (shape[0] / h)).astype(int)
        xs = (0.5 + np.arange(w) * (shape[1] / w)).astype(int)
        assert np.array_equal(o, a[np.ix_(ys, xs)]) and oi.nodata == 255.0
        assert oi.colormap[1].tolist() == [0, 0, 255]
    if shape[0] % 4 == 0 and shape[1] % 4 == 0:
        assert np.array_equal(geotiff.read_geotiff(p, overview=0)[0], a[::4, ::4])
    with pytest.raises(geotiff.GeoTiffError):
        geotiff.read_geotiff(p, overview=4)


@pytest.mark.parametrize('case', ['u8_palette', 'u16', 'i16', 'f32', 'f32_ragged'])
def test_written_files_open_in_an_independent_reader(tmp_path, case):
    """the writer is not only checked against this package's own reader.  Pillow
    (libtiff 4.x underneath: its own TIFF directory parser, DEFLATE, PREDICTOR=2 / 3 and tile assembly)
    opens every IFD -- full resolution and the four overviews -- of what `save_as_cog` here writes, and
    the pixels equal both the source array and what read_geotiff returns."""
    from PIL import Image, features
    if not features.check('libtiff'):
        pytest.skip('Pillow without libtiff')
    rng = np.random.default_rng(7)
    shape = (1300, 1100) if case != 'f32_ragged' else (1037, 515)
    kw = {}
    if case == 'u8_palette':
        a = rng.integers(0, 6, size=shape).astype(np.uint8)
        kw = dict(colormap={0: (255, 255, 255), 1: (0, 0, 255), 2: (0, 127, 255), 255: (0, 0, 0)}, nodata=255)
    elif case == 'u16':
        a = rng.integers(0, 11112, size=shape).astype(np.uint16)
        kw = dict(nodata=65535)
    elif case == 'i16':
        a = rng.integers(-9999, 12000, size=shape).astype(np.int16)
    else:
        a = rng.normal(800.0, 300.0, size=shape).astype(np.float32)
        a[40:60, 100:180] = np.nan
        a[0, 0], a[1, 1], a[2, 2] = np.inf, -0.0, 1e-42          # inf, signed zero, a denormal
        kw = dict(nodata=float('nan'))
    p = str(tmp_path / f'{case}.tif')
    geotiff.write_geotiff(p, a, overviews=geotiff.COG_OVERVIEW_FACTORS, metadata={'K': 'v'}, **kw)
    assert geotiff.validate_cog(p) == []
    with Image.open(p) as im:
        assert im.n_frames == 1 + len(geotiff.COG_OVERVIEW_FACTORS)
        assert im.tag_v2[317] == (3 if a.dtype.kind == 'f' else 2)      # PREDICTOR as core.py:66-69 asks
        assert im.tag_v2[259] == 8 and im.tag_v2[322] == 512            # DEFLATE, 512 x 512 tiles
        for k in range(im.n_frames):
            im.seek(k)
            theirs = np.array(im)
            ours, info = geotiff.read_geotiff(p, overview=None if k == 0 else k - 1)
            assert theirs.shape == ours.shape
            # (Pillow widens int16 to its 32-bit integer mode: compare values)
            assert np.array_equal(theirs.astype(np.int64) if theirs.dtype.kind in 'iu' else theirs,
                                  ours.astype(np.int64) if ours.dtype.kind in 'iu' else ours,
                                  equal_nan=a.dtype.kind == 'f'), (case, k)
            if k == 0:
                assert np.array_equal(ours, a, equal_nan=a.dtype.kind == 'f')
                assert ours.tobytes() == a.tobytes()                    # bit patterns: -0.0, NaN payload, denormal
            if case == 'u8_palette':
                assert im.mode == 'P' and im.getpalette()[3:6] == [0, 0, 255]


def test_float_predictor_and_foreign_layouts(tmp_path):
    """PREDICTOR=3 (TIFF Technical Note 3) round trip incl. float64, and a Pillow/libtiff-WRITTEN
    predictor-3 file read back by read_geotiff (VERDICT r01 'missing' 2: a GDAL-written Float32 DEM with
    PREDICTOR=3 could not even be read)."""
    from PIL import Image, features
    rng = np.random.default_rng(8)
    for dt in (np.float32, np.float64):
        a = rng.normal(size=(300, 257)).astype(dt)
        a[5, 5] = np.nan
        p = str(tmp_path / f'{np.dtype(dt).name}.tif')
        geotiff.write_geotiff(p, a, nodata=float('nan'))
        b, _ = geotiff.read_geotiff(p)
        assert b.dtype == a.dtype and b.tobytes() == a.tobytes()
    if not features.check('libtiff'):
        pytest.skip('Pillow without libtiff')
    a = rng.normal(100.0, 30.0, size=(200, 333)).astype(np.float32)
    q = str(tmp_path / 'foreign.tif')
    Image.fromarray(a).save(q, compression='tiff_adobe_deflate', tiffinfo={317: 3})
    with Image.open(q) as im:
        assert im.tag_v2[317] == 3                                      # libtiff really applied the predictor
    b, info = geotiff.read_geotiff(q)                                   # strips, written by another library
    assert b.dtype == np.float32 and np.array_equal(b, a)


def test_foreign_deflate_strips_chunky_and_short_last_strip(tmp_path):
    """Round 6: the reader inflates every block into one staging buffer (native codec, proteus_amd.codec) and undoes the
    predictor / moves the blocks with whole-array operations.  Files written by ANOTHER library (Pillow / libtiff) pin
    the layouts our own writer never produces: DEFLATE strips whose last strip is short, horizontal predictor on int16
    and on chunky RGB (three samples per pixel, differenced per sample), with both DEFLATE engines; and the codec's
    own contract (bound, thread count independence, corrupt / oversized streams)."""
    from PIL import Image, features
    from proteus_amd import codec
    if not features.check('libtiff'):
        pytest.skip('Pillow without libtiff')
    rng = np.random.default_rng(66)
    a16 = rng.integers(-3000, 12000, size=(203, 331)).astype(np.int16)
    rgb = rng.integers(0, 256, size=(157, 211, 3)).astype(np.uint8)
    u8 = (rng.integers(0, 5, size=(1000, 77)) * 50).astype(np.uint8)
    p16, prgb, pu8 = (str(tmp_path / n) for n in ('i16.tif', 'rgb.tif', 'u8.tif'))
    Image.fromarray(a16.view(np.uint16)).save(p16, compression='tiff_adobe_deflate', tiffinfo={317: 2})
    Image.fromarray(rgb).save(prgb, compression='tiff_adobe_deflate', tiffinfo={317: 2})
    Image.fromarray(u8).save(pu8, compression='tiff_adobe_deflate')
    for force in (False, True):
        codec.force_zlib(force)
        try:
            d = geotiff.open_geotiff(pu8)
            assert not d.tiled and d.down > 1 and d.info.height % d.bh           # several strips, the last one short
            assert np.array_equal(geotiff.read_geotiff(pu8)[0], u8)
            b16 = geotiff.read_geotiff(p16)[0]
            assert np.array_equal(b16.astype(np.uint16), a16.astype(np.uint16))  # ('I;16' is unsigned in the file)
            back, info = geotiff.read_geotiff(prgb)
            assert info.bands == 3 and np.array_equal(back, np.moveaxis(rgb, 2, 0))
            # our own files, every dtype, through this engine and back through Python's zlib
            for dt in (np.uint8, np.int16, np.uint16, np.float32):
                a = rng.integers(0, 200, size=(2, 700, 600)).astype(dt)
                p = str(tmp_path / f'own_{np.dtype(dt).name}_{int(force)}.tif')
                geotiff.write_geotiff(p, a, overviews=geotiff.COG_OVERVIEW_FACTORS if dt != np.float32 else None)
                assert np.array_equal(geotiff.read_geotiff(p)[0], a) and not geotiff.validate_cog(p)
                dd = geotiff.open_geotiff(p)
                import zlib
                raw = zlib.decompress(dd.buf[dd.offs[0]: dd.offs[0] + dd.cnts[0]])    # a standard zlib stream
                assert len(raw) == dd.block_bytes
        finally:
            codec.force_zlib(False)
    assert codec.engine() in ('libdeflate', 'zlib')
    # an UNCOMPRESSED file with a stray PREDICTOR tag: libtiff ignores the tag there (the predictor belongs to the codec)
    from PIL import TiffImagePlugin
    ifd = TiffImagePlugin.ImageFileDirectory_v2()
    ifd[317] = 2
    ifd[339] = 2
    s16 = rng.integers(-3000, 12000, size=(57, 91)).astype(np.int16)
    praw = str(tmp_path / 'raw_with_predictor_tag.tif')
    Image.fromarray(s16.view(np.uint16)).save(praw, tiffinfo=ifd)
    d = geotiff.open_geotiff(praw)
    assert d.comp == 1 and d.predictor == 1 and d.info.dtype == np.int16
    assert np.array_equal(geotiff.read_geotiff(praw)[0], s16)
    # the codec alone: the bytes do not depend on the thread count; errors are errors
    blocks = rng.integers(0, 4, size=(37, 4096)).astype(np.uint8)
    one = codec.deflate_uniform(blocks, 4096, 6, threads=1)
    many = codec.deflate_uniform(blocks, 4096, 6, threads=8)
    assert np.array_equal(one[2], many[2]) and all(
        one[0][o:o + k].tobytes() == many[0][o2:o2 + k].tobytes() for o, o2, k in zip(one[1], many[1], one[2]))
    blob = b''.join(one[0][o:o + k].tobytes() for o, k in zip(one[1], one[2]))
    offs = np.concatenate([[0], np.cumsum(one[2])[:-1]])
    dst = np.zeros_like(blocks)
    assert (codec.inflate_into(blob, offs, one[2], dst, 4096) == 4096).all() and np.array_equal(dst, blocks)
    with pytest.raises(codec.CodecError, match='more than'):
        codec.inflate_into(blob, offs, one[2], np.zeros((37, 100), np.uint8), 100)
    bad = bytearray(blob)
    bad[int(offs[5]) + 4] ^= 0xff
    with pytest.raises(codec.CodecError):
        codec.inflate_into(bytes(bad), offs, one[2], dst, 4096)
    with pytest.raises(codec.CodecError, match='outside'):
        codec.inflate_into(blob[:100], offs, one[2], dst, 4096)


def test_cubicspline_overviews_of_float_layers():
    """CUBICSPLINE overviews for non-integer layers (reference core.py:41-46).  GDAL is absent, so the
    kernel is pinned by its defining properties: weights sum to one (a constant stays constant), the cubic
    B-spline reproduces linear functions away from the border, NaN (the DEM's nodata) is excluded and
    renormalised, the levels are built in cascade, sizes are ceil(N / factor)."""
    const = np.full((400, 300), 123.25, np.float32)
    assert np.array_equal(geotiff.overview_cubicspline(const, 4), np.full((100, 75), 123.25, np.float32))
    yy, xx = np.mgrid[0:512, 0:640].astype(np.float64)
    ramp = (3.0 * xx - 2.0 * yy + 10.0).astype(np.float32)
    ov = geotiff.overview_cubicspline(ramp, 4)
    cy, cx = (np.arange(128) + 0.5) * 4 - 0.5, (np.arange(160) + 0.5) * 4 - 0.5   # source coordinates of the centres
    want = 3.0 * cx[None, :] - 2.0 * cy[:, None] + 10.0
    assert np.allclose(ov[3:-3, 3:-3], want[3:-3, 3:-3], rtol=0, atol=1e-3)
    holed = ramp.copy()
    holed[200:260, 300:380] = np.nan
    ovh = geotiff.overview_cubicspline(holed, 4)
    assert np.isnan(ovh[56, 84]) and not np.isnan(ovh).all()             # deep inside the hole: no support at all
    far = np.ones_like(ovh, bool)
    far[44:72, 68:102] = False
    assert np.allclose(ovh[far][np.isfinite(ov[far])], ov[far][np.isfinite(ov[far])], atol=1e-3)
    assert geotiff.overview_cubicspline(np.zeros((513, 41), np.float32), 4).shape == (129, 11)
    # the weights of one destination pixel: B-spline stretched by the ratio, normalised
    w = geotiff._bspline((np.arange(-8, 8) + 0.5) / 4.0)
    assert abs(w.sum() / 4.0 - 1.0) < 1e-12 and w[0] > 0 and geotiff._bspline(np.array([2.0, -2.5])).tolist() == [0.0, 0.0]


def test_writes_are_atomic(tmp_path, monkeypatch):
    """a writer killed mid-file must not leave a truncated product under the final name (the batch
    driver's --skip-existing keys on it)."""
    p = str(tmp_path / 'x.tif')
    geotiff.write_geotiff(p, np.zeros((10, 10), np.uint8))
    before = open(p, 'rb').read()
    real_replace = os.replace

    def boom(src, dst):
        raise RuntimeError('killed before the rename')
    monkeypatch.setattr(os, 'replace', boom)
    with pytest.raises(RuntimeError):
        geotiff.write_geotiff(p, np.ones((20, 20), np.uint8))
    with pytest.raises(RuntimeError):
        geotiff.write_png_palette(str(tmp_path / 'b.png'), np.zeros((4, 4), np.uint8), {0: (0, 0, 0)})
    monkeypatch.setattr(os, 'replace', real_replace)
    assert open(p, 'rb').read() == before and not os.path.exists(str(tmp_path / 'b.png'))


def test_multiband_product_has_the_reference_band_layout(tmp_path):
    """ADVICE r01 + r02: the reference creates ten Byte bands (:2663-2666) but its writing loop skips, WITHOUT
    advancing the band index, every name of band_description_dict that was not passed (:2673-2686), and the
    multi-band call never passes conf (:5383-5397).  Band k here is band k of a reference-made product:
    WTR, BWTR, DIAG, WTR-1, WTR-2, LAND, SHAD, CLOUD, DEM, and a tenth band that is never written.
    DIAG / DEM go through GDAL's Byte conversion; layers passed as None are nodata planes."""
    rng = np.random.default_rng(5)
    shape = (40, 50)
    u8 = lambda: rng.integers(0, 5, size=shape).astype(np.uint8)   # noqa: E731
    diag = rng.choice(np.array([0, 1, 11, 111, 10101, 11111, 65535], np.uint16), size=shape)
    dem = rng.normal(120.0, 200.0, size=shape).astype(np.float32)
    dem[0, 0] = np.nan
    layers = {'WTR': u8(), 'BWTR': u8(), 'DIAG': diag, 'WTR-1': u8(), 'WTR-2': u8(),
              'LAND': None, 'SHAD': rng.integers(0, 2, size=shape).astype(bool), 'CLOUD': u8(), 'DEM': dem}
    out = str(tmp_path / 'product.tif')
    D.save_dswx_product(layers, out, {'A': 'b'}, None)
    stack, info = geotiff.read_geotiff(out)
    assert list(D.band_description_dict) == ['WTR', 'BWTR', 'CONF', 'DIAG', 'WTR-1', 'WTR-2', 'LAND', 'SHAD',
                                             'CLOUD', 'DEM']
    written = ['WTR', 'BWTR', 'DIAG', 'WTR-1', 'WTR-2', 'LAND', 'SHAD', 'CLOUD', 'DEM']
    assert info.bands == 10 and stack.dtype == np.uint8 and info.nodata == 255.0
    for n in ('WTR', 'BWTR', 'WTR-1', 'WTR-2', 'CLOUD'):
        assert np.array_equal(stack[written.index(n)], layers[n]), n
    assert np.array_equal(stack[2], np.minimum(diag, 255).astype(np.uint8))            # saturated like GDT_Byte
    assert np.array_equal(stack[5], np.full(shape, 255, np.uint8))                     # LAND not produced
    assert np.array_equal(stack[6], layers['SHAD'].astype(np.uint8))
    want_dem = np.floor(np.clip(np.nan_to_num(dem.astype(np.float64), nan=0.0), 0, 255) + 0.5).astype(np.uint8)
    assert np.array_equal(stack[8], want_dem) and stack[8][0, 0] == 0
    assert not stack[9].any()                                                          # band 10: never written
    # the reference assigns `description` once and never resets it (:2686-2687): every written band carries WTR's
    assert info.descriptions == [D.band_description_dict['WTR']] * 9 + ['']
    # a CONF plane passed explicitly takes band 3, as the reference's loop would place it
    D.save_dswx_product(dict(layers, CONF=u8()), out, {'A': 'b'}, None)
    stack2, info2 = geotiff.read_geotiff(out)
    assert np.array_equal(stack2[3], stack[2]) and np.array_equal(stack2[9], stack[8])
    assert info2.descriptions == [D.band_description_dict['WTR']] * 10


def test_cog_validator_rejects_bad_layouts(tmp_path):
    a = np.zeros((600, 600), np.uint16)
    p = str(tmp_path / 'plain.tif')
    geotiff.write_geotiff(p, a)                       # no overviews: still a valid layout
    assert geotiff.validate_cog(p) == [] and len(geotiff.cog_layout(p)) == 1
    # move the main IFD pointer: a file whose first IFD is not at byte 8 must be flagged
    buf = bytearray(open(p, 'rb').read())
    (ifd,) = struct.unpack('<I', buf[4:8])
    (n,) = struct.unpack('<H', buf[ifd:ifd + 2])
    ifd_bytes = bytes(buf[ifd:ifd + 2 + 12 * n + 4])
    new_off = len(buf) + (len(buf) & 1)
    buf += b'\x00' * (new_off - len(buf)) + ifd_bytes
    buf[4:8] = struct.pack('<I', new_off)
    q = str(tmp_path / 'moved.tif')
    open(q, 'wb').write(bytes(buf))
    assert np.array_equal(geotiff.read_geotiff(q)[0], a)
    errs = geotiff.validate_cog(q)
    assert any('main IFD' in e for e in errs) and any('after its IFD' in e for e in errs)


def test_geotiff_reads_plain_strips(tmp_path):
    """A hand-made uncompressed, stripped, big-endian TIFF (what other writers produce)."""
    a = np.arange(6 * 5, dtype='>i2').reshape(6, 5)
    data = a.tobytes()
    entries = [(256, 3, 1, 5), (257, 3, 1, 6), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, 1),
               (273, 4, 1, 8), (277, 3, 1, 1), (278, 3, 1, 6), (279, 4, 1, len(data)), (339, 3, 1, 2)]
    ifd_off = 8 + len(data)
    buf = b'MM' + struct.pack('>HI', 42, ifd_off) + data + struct.pack('>H', len(entries))
    for tag, typ, cnt, val in entries:
        buf += struct.pack('>HHI', tag, typ, cnt) + (struct.pack('>HH', val, 0) if typ == 3
                                                      else struct.pack('>I', val))
    buf += struct.pack('>I', 0)
    p = tmp_path / 's.tif'
    p.write_bytes(buf)
    b, info = geotiff.read_geotiff(str(p))
    assert b.dtype == np.int16 and np.array_equal(b, a.astype(np.int16)) and info.nodata is None


def test_geotiff_rejects_garbage(tmp_path):
    p = tmp_path / 'bad.tif'
    p.write_bytes(b'not a tiff at all')
    with pytest.raises(geotiff.GeoTiffError):
        geotiff.read_geotiff(str(p))


# ---- runconfig ------------------------------------------------------------------------
def user_runconfig(tmp_path, **processing):
    doc = {'runconfig': {'name': 'x', 'groups': {
        'pge_name_group': {'pge_name': 'DSWX_HLS_PGE'},
        'input_file_group': {'input_file_path': ['a.B02.tif', 'a.B03.tif']},
        'dynamic_ancillary_file_group': {'dem_file_description': 'some DEM'},
        'primary_executable': {'product_type': 'DSWX_HLS'},
        'product_path_group': {'product_path': 'p', 'scratch_path': 'tmp', 'output_dir': 'out',
                               'product_id': 'PID', 'product_version': 0.5},
        'processing': processing,
        'browse_image_group': {'save_browse': True},
        'hls_thresholds': {'wigt': 0.2}}}}
    path = tmp_path / 'rc.yaml'
    path.write_text(yaml.safe_dump(doc))
    return str(path)


def test_default_runconfig_values():
    c = D.parse_runconfig_file()
    t = c.hls_thresholds
    assert (t.wigt, t.awgt, t.pswt_1_mndwi, t.pswt_1_nir, t.pswt_1_swir1, t.pswt_1_ndvi) == \
        (0.124, 0.0, -0.44, 1500, 900, 0.7)
    assert (t.pswt_2_mndwi, t.pswt_2_blue, t.pswt_2_nir, t.pswt_2_swir1, t.pswt_2_swir2,
            t.lcmask_nir) == (-0.5, 1000, 2500, 3000, 1000, 1200)
    assert c.mask_adjacent_to_cloud_mode == 'mask' and c.apply_aerosol_class_remapping is True
    assert c.aerosol_not_water_to_high_conf_water_fmask_values == [224, 160, 96]
    assert c.aerosol_partial_surface_aggressive_to_high_conf_water_fmask_values == \
        [224, 192, 160, 128, 96]
    assert c.shadow_masking_algorithm == 'sun_local_inc_angle' and c.apply_ocean_masking is False
    assert (c.min_slope_angle, c.max_sun_local_inc_angle) == (-5, 40)
    assert c.forest_mask_landcover_classes == [20, 50, 111, 113, 115, 116, 121, 123, 125, 126]
    assert (c.browse_image_height, c.not_water_in_browse, c.snow_in_browse) == (1024, 'white', 'cyan')


def test_runconfig_precedence_and_layer_names(tmp_path):
    path = user_runconfig(tmp_path, mask_adjacent_to_cloud_mode='ignore', save_conf=False)
    args = D.get_dswx_hls_cli_parser().parse_args(
        [path, '--wtr', 'my_wtr.tif', '--max-sun-local-inc-angle', '33'])
    c = D.parse_runconfig_file(path, args)
    assert c.hls_thresholds.wigt == 0.2 and c.hls_thresholds.awgt == 0.0      # user over default
    assert args.mask_adjacent_to_cloud_mode == 'ignore'                        # user runconfig
    assert args.max_sun_local_inc_angle == 33.0                                # CLI wins
    assert args.output_interpreted_band == 'my_wtr.tif'                        # CLI wins
    assert args.output_binary_water == os.path.join('out', 'PID_v0.5_B02_BWTR.tif')
    assert args.output_confidence_layer is None                                # save_conf False
    assert args.output_diagnostic_layer == os.path.join('out', 'PID_v0.5_B04_DIAG.tif')
    assert args.output_non_masked_dswx == os.path.join('out', 'PID_v0.5_B05_WTR-1.tif')
    assert args.output_shadow_masked_dswx == os.path.join('out', 'PID_v0.5_B06_WTR-2.tif')
    assert args.output_cloud_layer == os.path.join('out', 'PID_v0.5_B09_CLOUD.tif')
    assert args.output_dem_layer == os.path.join('out', 'PID_v0.5_B10_DEM.tif')
    assert args.output_rgb_file is None                                        # save_rgb default False
    assert args.output_browse_image == os.path.join('out', 'PID_v0.5_BROWSE.png')
    assert args.input_list == ['a.B02.tif', 'a.B03.tif']
    assert (args.product_id, args.product_version, args.scratch_dir) == ('PID', '0.5', 'tmp')
    assert args.dem_file_description == 'some DEM'


def test_runconfig_schema_violations(tmp_path):
    for bad in (dict(mask_adjacent_to_cloud_mode='bogus'), dict(min_slope_angle=500),
                dict(save_wtr='yes'), dict(unknown_key=1),
                dict(aerosol_not_water_to_high_conf_water_fmask_values=[1.5])):
        with pytest.raises(rc.RunconfigError):
            D.parse_runconfig_file(user_runconfig(tmp_path, **bad))
    with pytest.raises(Exception, match='ERROR invalid file'):
        D.parse_runconfig_file(str(tmp_path / 'missing.yaml'))


def test_deep_update_keeps_defaults_on_none():
    assert rc.deep_update({'a': {'b': 1, 'c': 2}}, {'a': {'b': None, 'c': 3}, 'd': None}) == \
        {'a': {'b': 1, 'c': 3}}


def test_cli_accepts_every_reference_flag():
    flags = ['--dem', 'd', '--dem-description', 'x', '-c', 'l', '--landcover-description', 'x',
             '-w', 'w', '--worldcover-description', 'x', '-s', 's', '--shoreline-shape-description',
             'x', '-o', 'o', '--interpreted-band', 'o', '--output-rgb', 'o', '--output-infrared-rgb',
             'o', '--bwtr', 'o', '--conf', 'o', '--diag', 'o', '--wtr-1', 'o', '--wtr-2', 'o',
             '--land', 'o', '--shad', 'o', '--cloud', 'o', '--out-dem', 'o', '--browse', 'o',
             '--bheight', '10', '--bwidth', '10', '--exclude-psw-aggressive-in-browse',
             '--not-water-in-browse', 'nodata', '--cloud-in-browse', 'gray', '--snow-in-browse',
             'cyan', '--offset-and-scale-inputs', '--temp-dir', 't', '--pid', 'p',
             '--product-version', '1', '--check-ancillary-inputs-coverage', '--apply-ocean-masking',
             '--apply-aerosol-masking', '--shadow-masking-algorithm', 'otsu', '--min-slope-angle',
             '-5', '--max-sun-local-inc-angle', '40', '--mask-adjacent-to-cloud-mode', 'cover',
             '--ocean-masking-distance-km', '1', '--debug', '--log', 'l', '--full-log-format']
    a = D.get_dswx_hls_cli_parser().parse_args(['in.tif'] + flags)
    assert a.output_binary_water == 'o' and a.browse_image_height == 10 and a.flag_debug
    assert a.mask_adjacent_to_cloud_mode == 'cover' and a.apply_ocean_masking is True
    # the reference's accidental concatenated spellings parse too
    a = D.get_dswx_hls_cli_parser().parse_args(['in.tif', '--bwtr--output-binary-water', 'q'])
    assert a.output_binary_water == 'q'


# ---- tables / metadata -----------------------------------------------------------------
def test_tables_match_oracle():
    from oracle import dswx_oracle as o
    assert D.interpreted_dswx_band_dict == o.DIAG_TO_CLASS
    assert D.collapse_wtr_classes_dict == o.COLLAPSE
    assert list(D.layer_names_to_args_dict)[:10] == list(D.band_description_dict)
    assert D.AEROSOL_REMAPPING_MAX_NIR == 1000.0


def test_hls_metadata_harvest():
    md = {}
    ok = D._harvest_hls_metadata({'SENSOR': 'OLI_TIRS; OLI_TIRS', 'MEAN_SUN_AZIMUTH_ANGLE': '1',
                                  'LANDSAT_PRODUCT_ID': 'LC09_L1TP_x', 'cloud_coverage': '7',
                                  'SENSING_TIME': 't'}, md)
    assert ok and md['SPACECRAFT_NAME'] == 'Landsat-9' and md['SENSOR'] == 'OLI'
    assert md['INPUT_HLS_PRODUCT_CLOUD_COVERAGE'] == '7' and md['SENSOR_PRODUCT_ID'] == 'LC09_L1TP_x'
    md = {}
    assert D._harvest_hls_metadata({'SPACECRAFT_NAME': 'Sentinel-2B', 'PRODUCT_URI': 'S2B'}, md)
    assert md['SENSOR'] == 'MSI' and md['SENSOR_PRODUCT_ID'] == 'S2B'
    assert not D._harvest_hls_metadata({'SPACECRAFT_NAME': 'Terra'}, {})
    assert not D._harvest_hls_metadata({'SENSOR': 'MODIS'}, {})
    assert not D._harvest_hls_metadata({}, {})


def test_metadata_dicts():
    md = D._get_dswx_metadata_dict('PID', None)
    assert md['PRODUCT_VERSION'] == D.SOFTWARE_VERSION and md['PROJECT'] == 'OPERA'
    md['SPACECRAFT_NAME'] = 'Sentinel-2A'
    D._populate_dswx_metadata_datasets(md, 'HLS.S30.x', dem_file='/a/b/dem.tif',
                                       landcover_file_description='CGLS')
    assert md['DEM_SOURCE'] == 'dem.tif' and md['LANDCOVER_SOURCE'] == 'CGLS'
    assert md['WORLDCOVER_SOURCE'] == 'NOT_PROVIDED'
    assert md['SHORELINE_SOURCE'] == 'NOT_PROVIDED_OR_NOT_USED'
    assert 'Copernicus Sentinel' in md['LICENSE'] and 'Copernicus programme' in md['LICENSE']
    D._populate_dswx_metadata_processing_parameters(
        md, False, True, [[224, 160, 96]] * 2 + [[224, 192]] * 2, 'sun_local_inc_angle', -5, 40,
        'mask', [20, 50], 1)
    assert md['AEROSOL_CLASS_REMAPPING_ENABLED'] == 'TRUE'
    assert md['AEROSOL_NOT_WATER_TO_HIGH_CONF_WATER_FMASK_VALUES'] == '224,160,96'
    assert md['SHADOW_MASKING_ALGORITHM'] == 'SUN_LOCAL_INC_ANGLE' and md['MIN_SLOPE_ANGLE'] == -5
    assert md['OCEAN_MASKING_ENABLED'] == 'FALSE'
    assert md['OCEAN_MASKING_SHORELINE_DISTANCE_KM'] == 'NOT_USED'
    assert md['FOREST_MASK_LANDCOVER_CLASSES'] == '20,50'


def test_confidence_colour_table():
    ct = D._get_confidence_layer_ctable()
    assert ct[10] == D.get_transparency_rgb_vals((175, 175, 175), (255, 255, 255), 0.52)
    assert ct[21] == (0, 255, 255) and ct[252] == (0, 0, 0) and ct[254] == (0, 0, 127)
    with pytest.raises(ValueError):
        D.get_transparency_rgb_vals((0, 0, 0), (1, 1, 1), 1.5)


# ---- error conventions of generate_dswx_layers (no GPU is reached) ----------------------
def test_generate_rejects_bad_parameters(tmp_path):
    rcfile, files, _, _ = synth_hls.make(str(tmp_path), size=64)
    with pytest.raises(ValueError, match='Invalid shadow masking algorithm'):
        D.generate_dswx_layers(files, shadow_masking_algorithm='magic')
    with pytest.raises(Exception, match='ERROR mask adjacent to cloud/cloud-shadow mode'):
        D.generate_dswx_layers(files, mask_adjacent_to_cloud_mode='bogus')
    # what still needs GDAL is refused before anything is loaded
    with pytest.raises(NotImplementedError, match='GDAL'):
        D.generate_dswx_layers(files, shoreline_shapefile='coast.shp', apply_ocean_masking=True)
    with pytest.raises(NotImplementedError, match='otsu'):
        D.generate_dswx_layers(files, dem_file='dem.tif', shadow_masking_algorithm='otsu')


def test_grid_margin_detection():
    """`_grid_margin`: is an ancillary raster already on the product grid (what gdal.Warp would
    have produced), and with which margin."""
    gt = (600000.0, 30.0, 0.0, 4000020.0, 0.0, -30.0)

    def info(gtx, h, w):
        i = geotiff.GeoTiffInfo()
        i.geo_tags = geotiff.geo_tags_from_geotransform(gtx)
        i.height, i.width = h, w
        return i
    assert D._grid_margin(info(gt, 100, 120), gt, 100, 120) == 0
    m50 = (gt[0] - 50 * 30, 30.0, 0.0, gt[3] + 50 * 30, 0.0, -30.0)
    assert D._grid_margin(info(m50, 200, 220), gt, 100, 120) == 50
    assert D._grid_margin(info(m50, 200, 221), gt, 100, 120) is None          # asymmetric cover
    assert D._grid_margin(info((gt[0] + 7, 30.0, 0.0, gt[3], 0.0, -30.0), 100, 120), gt, 100, 120) is None
    assert D._grid_margin(info((gt[0], 10.0, 0.0, gt[3], 0.0, -10.0), 300, 360), gt, 100, 120) is None
    assert D._grid_margin(info((gt[0], 10.0, 0.0, gt[3], 0.0, -10.0), 300, 360), gt, 100, 120, scale=3) == 0
    assert D._grid_margin(info((gt[0] + 30, 30.0, 0.0, gt[3], 0.0, -30.0), 100, 120), gt, 100, 120) is None   # inside


def test_generate_returns_false_on_unreadable_input(tmp_path):
    assert D.generate_dswx_layers([str(tmp_path / 'nope.B02.tif'),
                                   str(tmp_path / 'nope.Fmask.tif')]) is False
    _, files, _, _ = synth_hls.make(str(tmp_path), size=64)
    assert D.generate_dswx_layers(files[:3]) is False          # bands missing


def test_compare_products(tmp_path, capsys):
    a = np.arange(100, dtype=np.uint8).reshape(10, 10)
    geo = geotiff.geo_tags_from_geotransform((0, 30, 0, 0, 0, -30))
    md = {'PRODUCT_ID': 'x', 'PROCESSING_DATETIME': 't1', 'LICENSE': 'l1'}
    f1, f2, f3, f4 = (str(tmp_path / n) for n in ('1.tif', '2.tif', '3.tif', '4.tif'))
    geotiff.write_geotiff(f1, a, geo_tags=geo, metadata=md)
    geotiff.write_geotiff(f2, a, geo_tags=geo, metadata=dict(md, PROCESSING_DATETIME='t2', LICENSE='z'))
    b = a.copy()
    b[3, 4] += 1
    geotiff.write_geotiff(f3, b, geo_tags=geo, metadata=md)
    geotiff.write_geotiff(f4, a, geo_tags=geo, metadata=dict(md, PRODUCT_ID='y'))
    assert D.compare_dswx_hls_products(f1, f2) is True
    assert D.compare_dswx_hls_products(f1, f3) is False
    assert '(x: 4, y: 3)' in capsys.readouterr().out
    assert D.compare_dswx_hls_products(f1, f4) is False
    assert D.compare_dswx_hls_products(f1, str(tmp_path / 'missing.tif')) is False


def test_resident_batch_layout_rule():
    """dswx_batch_layout (pure function of the C-ABI, no device): where dswx_batch_create puts every plane.
    256-byte aligned offsets, no overlap, tile stride padded to 256 px by default, inputs then outputs, counters last;
    with SEPARATE_OUTPUTS the arena holds the inputs only."""
    from proteus_amd import _capi
    T = 3660
    lay = _capi.batch_layout(256, T, T)
    assert lay['tile_stride'] == 13395712 and lay['tile_stride'] % 256 == 0 and lay['tile_stride'] >= T * T
    px = 256 * lay['tile_stride']
    order = sorted(lay['planes'], key=lambda n: lay['planes'][n][0])
    assert order == list(_capi.BAND_NAMES) + ['fmask', 'diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud', 'counters']
    cur = 0
    for name in order:
        off, nbytes = lay['planes'][name]
        assert off == cur and off % 256 == 0, name
        want = 256 * 24 if name == 'counters' else px * (2 if name in _capi.BAND_NAMES + ('diag',) else 1)
        assert nbytes == (want + 255) // 256 * 256, name
        cur = off + nbytes
    assert lay['arena_bytes'] == cur == 21 * px + 256 * 24               # 13 + 8 bytes per pixel
    assert lay['write_span_bytes'] == 8 * px
    # the optional planes, an explicit (contiguous) stride, a ragged size
    lay = _capi.batch_layout(3, 7, 9, masks=True, extra_layers=('wtr1_aerosol', 'browse'), tile_stride=63)
    assert lay['tile_stride'] == 63 and set(lay['planes']) == set(_capi.PLANE_INDEX)
    spans = sorted(lay['planes'].values())
    assert all(a[0] + a[1] <= b[0] for a, b in zip(spans, spans[1:]))   # no overlap
    assert all(off % 256 == 0 for off, _ in spans)
    # one allocation per output plane: the arena is the inputs' (+ counters)
    sep = _capi.batch_layout(256, T, T, separate_outputs=True)
    assert sep['arena_bytes'] == 13 * px + 256 * 24 and sep['write_span_bytes'] == 0
    assert all(sep['planes'][n][0] == 0 for n in ('diag', 'wtr1', 'cloud'))
    # errors
    with pytest.raises(_capi.DswxError):
        _capi.batch_layout(2, 8, 8, tile_stride=63)                        # smaller than the tile
    with pytest.raises(_capi.DswxError):
        _capi.batch_layout(-1, 8, 8)
    with pytest.raises(ValueError):
        _capi.batch_layout(1, 8, 8, extra_layers=('mndwi',))
    lib = _capi.load_library()
    lay_c, geom = _capi.BatchLayout(), _capi.BatchGeom(1, 8, 8, 0)
    assert lib.dswx_batch_layout(ctypes.byref(geom), 1 << 20, ctypes.byref(lay_c)) == _capi.ERR_ARG   # unknown flag
    assert b'unknown batch flag' in lib.dswx_last_error()


def test_scale_report_reads_bench_lines(tmp_path):
    """tools/scale_report.py: the table north_star asks for, from bench.py lines of several GPU counts (here the two
    round-5 lines that exist: N = 1, and the two-rank rehearsal that shares ONE device -- its efficiency is ~0.5 by
    construction and the note says why)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, 'profiles', n) for n in ('r05_bench_driver_style_final.json',
                                                         'r05_rehearsal_driver_command_2ranks_one_device.json')]
    res = subprocess.run([sys.executable, os.path.join(root, 'tools', 'scale_report.py')] + files, capture_output=True,
                         text=True, timeout=60)
    assert res.returncode == 0, res.stderr
    rep = json.loads(res.stdout)
    r1, r2 = rep['rows']
    assert r1['ranks'] == 1 and r1['efficiency_vs_n1'] == 1.0 and rep['baseline_n1_Mpx_s'] == r1['value_Mpx_s']
    assert r2['ranks'] == 2 and r2['distinct_gpus'] == 1 and 0.4 < r2['efficiency_vs_n1'] < 0.6 and r2['error'] is None
    assert r2['strong_4096_tiles']['parity'] == 'bit-exact' and 'distinct device' in r2['note']


def test_stage_clock_reports_work_and_wall_per_stage():
    """proteus_amd.stages (round 6: where the wall time of a product run goes): spans from several threads, per stage
    the sum of the spans (work) and the length of their UNION (wall); off unless started."""
    import threading
    import time
    from proteus_amd import stages
    assert not stages.recording()
    with stages.span('ignored'):
        pass
    stages.start()
    t0 = time.perf_counter()
    stages.add('a', t0, t0 + 1.0)
    stages.add('a', t0 + 0.5, t0 + 2.0)           # overlaps the first: union 2.0, work 2.5
    stages.add('a', t0 + 3.0, t0 + 3.5)           # disjoint
    stages.add('b', t0 + 1.0, t0 + 1.25)

    def worker():
        with stages.span('c'):
            time.sleep(0.01)
    ts = [threading.Thread(target=worker) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    rep = stages.stop()
    assert not stages.recording()
    a = rep['stages']['a']
    assert a['spans'] == 3 and abs(a['thread_s'] - 3.0) < 1e-3 and abs(a['wall_s'] - 2.5) < 1e-3
    assert rep['stages']['b'] == {'spans': 1, 'thread_s': 0.25, 'wall_s': 0.25}
    assert rep['stages']['c']['spans'] == 4 and rep['stages']['c']['thread_s'] >= 0.04
    assert list(rep['stages']) == ['a', 'c', 'b'] and rep['wall_s'] >= 3.5          # stages in the order of their first start
    assert stages.report([]) == {'wall_s': 0.0, 'stages': {}}


@pytest.mark.parametrize('shape', [(3660, 3660), (1, 1), (1, 7), (513, 1025), (4096, 512), (29, 58)])
def test_cog_layout_is_the_host_writers_layout(shape):
    """dswx_cog_layout (ABI v6, no device needed) against the levels the host writer makes: sizes ceil(N / f), blocks,
    offsets in file-independent level order, for u8 / u16 / f32 and the reference's factors (core.py:37)."""
    from proteus_amd import _capi
    h, w = shape
    for itemsize, factors in ((1, geotiff.COG_OVERVIEW_FACTORS), (2, geotiff.COG_OVERVIEW_FACTORS), (4, ()), (1, (1, 4))):
        lay = _capi.cog_layout(h, w, itemsize, factors, 512)
        arr = np.zeros(shape, {1: np.uint8, 2: np.uint16, 4: np.float32}[itemsize])
        levels = [arr] + [geotiff.overview_nearest(arr, f) for f in factors if f > 1 and shape != (1, 1)]
        assert lay['n_levels'] == len(levels)
        off = 0
        for lv, want in zip(lay['levels'], levels):
            b = geotiff.blocked_level(want[None], 512, 2 if itemsize < 4 else 3)
            assert (lv['height'], lv['width'], lv['blocks_down'], lv['blocks_across']) == (b.height, b.width, b.down, b.across)
            assert lv['offset_bytes'] == off
            off += b.n_blocks * b.block_bytes
        assert lay['total_bytes'] == off
    with pytest.raises(_capi.DswxError):
        _capi.cog_layout(h, w, 3, (), 512)
    with pytest.raises(_capi.DswxError):
        _capi.cog_layout(h, w, 1, (), 100)


def test_reader_survives_damaged_files(tmp_path):
    """A damaged band file is an ERROR the product reports (generate_dswx_layers returns False after 'ERROR could not open',
    as the reference does when gdal.Open fails, dswx_hls.py:4988-4990) -- never a crash of the native codec, a read outside
    the file or an absurd allocation: 400 mutations (byte flips in the header and directory, in the block table, in the
    compressed data; truncations; a few bytes of garbage) of tiled, stripped, multi-level and Float32 files either decode
    or raise GeoTiffError -- the one error type the loader turns into 'ERROR could not open'."""
    rng = np.random.default_rng(99)
    a = rng.integers(0, 3000, size=(300, 260)).astype(np.int16)
    files = []
    p = str(tmp_path / 'tiled.tif')
    geotiff.write_geotiff(p, a, tile=64, overviews=(2, 4), nodata=-9999, metadata={'K': 'v'})
    files.append(p)
    p = str(tmp_path / 'float.tif')
    geotiff.write_geotiff(p, a.astype(np.float32) / 7, tile=128)
    files.append(p)
    p = str(tmp_path / 'raw.tif')
    geotiff.write_geotiff(p, a.astype(np.uint8), compress=False, tile=256)
    files.append(p)
    ok = bad = 0
    for case in range(400):
        src = open(files[case % len(files)], 'rb').read()
        buf = bytearray(src)
        kind = case % 5
        if kind == 0:                                   # header + first directory
            for _ in range(int(rng.integers(1, 4))):
                buf[int(rng.integers(0, min(400, len(buf))))] = int(rng.integers(0, 256))
        elif kind == 1:                                 # anywhere
            for _ in range(int(rng.integers(1, 20))):
                buf[int(rng.integers(0, len(buf)))] ^= int(rng.integers(1, 256))
        elif kind == 2:                                 # truncated
            buf = buf[:int(rng.integers(0, len(buf)))]
        elif kind == 3:                                 # the tail (block data) overwritten with noise
            k = int(rng.integers(1, max(2, len(buf) // 3)))
            buf[-k:] = rng.integers(0, 256, size=k, dtype=np.uint8).tobytes()
        else:                                           # a 32-bit field set to a huge value somewhere in the directory
            q = int(rng.integers(8, min(600, len(buf) - 4)))
            buf[q:q + 4] = (0xfffffff0).to_bytes(4, 'little')
        q = str(tmp_path / 'damaged.tif')
        open(q, 'wb').write(bytes(buf))
        try:
            for ov in (None, 0):
                arr, info = geotiff.read_geotiff(q, overview=ov)
                assert arr.size < 50_000_000
            ok += 1
        except geotiff.GeoTiffError:                    # the ONE error type a damaged file may raise
            bad += 1
    assert ok + bad == 400 and bad > 100


def test_every_tool_and_helper_script_parses():
    """tools/, tools/lab/, tests/helpers/, bin/ hold the measurement and soak scripts the profiles were made with: they only
    run on a GPU box, so at least their syntax is checked here."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sorted(glob.glob(os.path.join(root, 'tools', '*.py')) + glob.glob(os.path.join(root, 'tools', 'lab', '*.py')) +
                   glob.glob(os.path.join(root, 'tests', 'helpers', '*.py')) + glob.glob(os.path.join(root, 'bin', '*.py')) +
                   [os.path.join(root, 'bench.py'), os.path.join(root, '__graft_entry__.py')])
    assert len(paths) > 20
    for p in paths:
        ast.parse(open(p).read(), filename=p)


def test_lzw_files_read_like_deflate_files(tmp_path):
    """TIFF compression 5 (LZW): ancillary rasters from other GDAL tools are often LZW files, and `gdal.Open` reads them like
    any other.  libtiff (through Pillow) writes them -- strips, predictor 1 / 2 / 3, one- to four-byte samples, rasters that
    compress 150 : 1 and rasters that do not compress -- and the reader (native decoder, proteus_amd/csrc/dswx_codec.cpp)
    returns the arrays; a damaged file is an unreadable file, not a crash.  (The decoder is also checked against an
    encoder of the sanitizer harness's own: tests/native/codec_stress.cpp.)"""
    from PIL import Image, features
    if not features.check('libtiff'):
        pytest.skip('Pillow without libtiff')
    rng = np.random.default_rng(55)
    cases = (('classes', rng.integers(0, 5, size=(700, 513)).astype(np.uint8), 1),
             ('bytes_p2', rng.integers(0, 255, size=(300, 1000)).astype(np.uint8), 2),
             ('flat', np.full((2000, 1500), 7, np.uint8), 1),
             ('u16_p2', rng.integers(0, 9000, size=(333, 1000)).astype(np.uint16), 2),
             ('i32', rng.integers(-10 ** 6, 10 ** 6, size=(100, 77)).astype(np.int32), 1),
             ('f32', rng.normal(size=(300, 257)).astype(np.float32), 1),
             ('f32_p3', (rng.normal(size=(300, 257)) * 100).astype(np.float32), 3),
             ('noise', rng.integers(0, 256, size=(1024, 1024)).astype(np.uint8), 1),
             ('one', np.array([[9]], np.uint8), 1))
    for name, arr, pred in cases:
        p = str(tmp_path / f'{name}.tif')
        Image.fromarray(arr).save(p, compression='tiff_lzw', tiffinfo={317: pred} if pred > 1 else {})
        d = geotiff.open_geotiff(p)
        assert d.comp == 5 and d.predictor == pred, name
        got, info = geotiff.read_geotiff(p)
        assert got.dtype == arr.dtype and np.array_equal(got, arr), name
    # damage: flipped bytes inside the strips, a truncated file
    p = str(tmp_path / 'classes.tif')
    raw = bytearray(open(p, 'rb').read())
    d = geotiff.open_geotiff(p)
    for trial in range(40):
        bad = bytearray(raw)
        for _ in range(6):
            i = int(rng.integers(0, d.n_blocks))
            bad[d.offs[i] + int(rng.integers(0, d.cnts[i]))] ^= int(rng.integers(1, 256))
        q = str(tmp_path / 'bad.tif')
        open(q, 'wb').write(bytes(bad))
        try:
            got, _ = geotiff.read_geotiff(q)
            assert got.shape == (700, 513)
        except geotiff.GeoTiffError:
            pass
    open(q, 'wb').write(bytes(raw[:len(raw) // 2]))
    with pytest.raises(geotiff.GeoTiffError):
        geotiff.read_geotiff(q)
