"""oracle/cog_oracle.py (the row-by-row restatement of the raster-format steps of SURVEY section 8 f4) pinned against an
independent implementation -- libtiff, through Pillow -- and used as the checker of the product's host writer / reader
(proteus_amd/geotiff.py) on the CPU; tests/test_gpu_writer.py uses it as the checker of the device kernels."""
import zlib

import numpy as np
import pytest

from oracle import cog_oracle as co
from proteus_amd import geotiff

FACTORS = geotiff.COG_OVERVIEW_FACTORS


def _arr(rng, shape, dtype):
    if np.dtype(dtype).kind == 'f':
        a = rng.normal(100.0, 40.0, size=shape).astype(dtype)
        a[rng.random(shape) < 0.05] = np.nan
        return a
    info = np.iinfo(dtype)
    return rng.integers(info.min, info.max + 1, size=shape).astype(dtype)


@pytest.mark.parametrize('dtype', [np.uint8, np.uint16, np.int16, np.float32])
def test_oracle_blocks_decode_in_libtiff(tmp_path, dtype):
    """ENCODE side pinned: a TIFF whose block bytes are the ORACLE's (tiling, zero padding, horDiff / fpDiff) -- only the
    directory and DEFLATE come from the product's writer -- is decoded by libtiff to the array, every overview level to
    the oracle's NEAREST pick of it."""
    from PIL import Image, features
    if not features.check('libtiff'):
        pytest.skip('Pillow without libtiff')
    rng = np.random.default_rng(3)
    arr = _arr(rng, (150, 201), dtype)
    predictor = 3 if np.dtype(dtype).kind == 'f' else 2
    factors = () if predictor == 3 else (4, 16)
    levels = [geotiff.BlockedLevel(h, w, 1, dtype, 64, predictor, data) for h, w, data in co.cog_levels(arr, factors, 64, predictor)]
    p = str(tmp_path / 'oracle_blocks.tif')
    geotiff.write_geotiff(p, None, levels=levels)
    rasters = [arr] + [co.nearest_overview(arr, f) for f in factors]
    with Image.open(p) as im:
        assert im.n_frames == len(rasters)
        for k, want in enumerate(rasters):
            im.seek(k)
            theirs = np.array(im)
            assert theirs.shape == want.shape
            if predictor == 3:
                assert theirs.astype(np.float32).tobytes() == want.tobytes()
            else:
                assert np.array_equal(theirs.astype(np.int64), want.astype(np.int64)), (dtype, k)


@pytest.mark.parametrize('dtype', [np.uint8, np.int16, np.float32])
def test_oracle_decodes_blocks_written_by_libtiff(tmp_path, dtype):
    """DECODE side pinned: the strips of a file Pillow / libtiff wrote (DEFLATE, PREDICTOR 2 / 3; the last strip short),
    inflated with Python's zlib, through the oracle's horAcc / fpAcc -> the array."""
    from PIL import Image, TiffImagePlugin, features
    if not features.check('libtiff'):
        pytest.skip('Pillow without libtiff')
    rng = np.random.default_rng(4)
    arr = _arr(rng, (1201, 57), dtype)
    predictor = 3 if np.dtype(dtype).kind == 'f' else 2
    ifd = TiffImagePlugin.ImageFileDirectory_v2()
    ifd[317] = predictor
    if dtype == np.int16:
        ifd[339] = 2
    p = str(tmp_path / 'libtiff.tif')
    Image.fromarray(arr.view(np.uint16) if dtype == np.int16 else arr).save(p, compression='tiff_adobe_deflate', tiffinfo=ifd)
    d = geotiff.open_geotiff(p)
    assert not d.tiled and d.down > 1 and d.predictor == predictor
    raw = bytearray(d.n_blocks * d.block_bytes)
    for i in range(d.n_blocks):
        blk = zlib.decompress(d.buf[d.offs[i]: d.offs[i] + d.cnts[i]])
        raw[i * d.block_bytes: i * d.block_bytes + len(blk)] = blk
    got = co.unblocks(np.frombuffer(bytes(raw), np.uint8), dtype, d.info.height, d.info.width, d.bw, d.bh, predictor)
    assert got.tobytes() == arr.tobytes()


@pytest.mark.parametrize('shape', [(1, 1), (7, 3), (65, 130), (150, 201)])
@pytest.mark.parametrize('dtype', [np.uint8, np.uint16, np.float32])
def test_host_writer_and_reader_against_the_oracle(shape, dtype):
    """The product's whole-array forms (geotiff.blocked_level, overview_nearest, TiffDirectory.untile's arithmetic)
    against the row-by-row oracle: block bytes of every level, and the way back."""
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    arr = _arr(rng, shape, dtype)
    predictor = 3 if np.dtype(dtype).kind == 'f' else 2
    for tile in (16, 64):
        factors = () if predictor == 3 else FACTORS
        want = co.cog_levels(arr, factors, tile, predictor)
        rasters = [arr] + [geotiff.overview_nearest(arr, f) for f in factors if shape != (1, 1)]
        assert len(want) == len(rasters)
        for (h, w, data), r in zip(want, rasters):
            lv = geotiff.blocked_level(r[None], tile, predictor)
            assert (lv.height, lv.width) == (h, w)
            assert np.array_equal(np.asarray(lv.data).reshape(-1).view(np.uint8), data), (shape, dtype, tile)
            back = co.unblocks(data, dtype, h, w, tile, tile, predictor)
            assert back.tobytes() == np.ascontiguousarray(r).tobytes()
    if predictor == 2:          # no predictor at all
        lv = geotiff.blocked_level(arr[None], 16, 1)
        assert np.array_equal(np.asarray(lv.data).reshape(-1).view(np.uint8), co.blocks(arr, 16, 1))


def test_rgb_oracle_statement():
    rng = np.random.default_rng(5)
    bands = [rng.integers(-100, 9000, size=(40, 50)).astype(np.int16) for _ in range(3)]
    diag = rng.integers(0, 11112, size=(40, 50)).astype(np.uint16)
    diag[::7, ::3] = 65535
    out = co.rgb_planes(bands, diag, [1e-4, 2e-4, 1.0], [0.0, -3.0, 12.5])
    assert out.dtype == np.float32 and out.shape == (3, 40, 50)
    assert np.isnan(out[:, diag == 65535]).all() and not np.isnan(out[:, diag != 65535]).any()
    assert out[1, 1, 1] == np.float32(2e-4) * (np.float32(max(int(bands[1][1, 1]), 1)) - np.float32(-3.0))


@pytest.mark.parametrize('shape', [(37, 53), (64, 64), (5, 3), (1, 7), (130, 90), (2, 2)])
def test_cubicspline_byte_and_resize_statements_against_the_oracle(shape):
    """The host writer's whole-array statements of the three GDAL rules round 6 moved onto the device -- CUBICSPLINE
    overview pyramid of a Float32 layer (core.py:41-46), the Byte bands of the multi-band file (dswx_hls.py:2663-2666), the
    browse image's resize (:5335-5349) -- against oracle/cog_oracle.py's element-by-element restatements: bit for bit, NaN
    holes, infinities and rounding edges included; plus the properties that define the convolution (a constant raster
    stays constant, weights renormalised around NaN, all-NaN support -> NaN)."""
    from oracle import cog_oracle as co
    from proteus_amd import dswx_hls as D
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    a = (rng.normal(size=shape) * 1000).astype(np.float32)
    a[rng.random(shape) < 0.05] = np.nan
    if shape[0] > 30:
        a[3:22, 4:30] = np.nan
        a[25, 7], a[27, 9] = np.inf, -np.inf
    want = co.cubicspline_pyramid(a, geotiff.COG_OVERVIEW_FACTORS)
    got, prev = [a], 1
    for f in geotiff.COG_OVERVIEW_FACTORS:                     # write_geotiff's cascade
        lv = geotiff.overview_cubicspline(got[-1], f // prev) if prev > 1 and f % prev == 0 else geotiff.overview_cubicspline(a, f)
        if lv.shape != tuple(-(-n // f) for n in shape):
            lv = geotiff.overview_cubicspline(a, f)
        got.append(lv)
        prev = f
    assert len(got) == len(want) == 5
    for k, (g, w) in enumerate(zip(got, want)):
        assert g.dtype == np.float32 and g.shape == w.shape and np.array_equal(g, w, equal_nan=True), k
    if shape[0] > 30:
        assert np.isnan(want[1]).any() and np.isinf(want[1]).any()
    flat = np.full(shape, 3.25, np.float32)
    flat[0, 0] = np.nan                                         # renormalised: the constant survives a missing tap
    lv = co.cubicspline_overview(flat, 4)
    assert np.allclose(lv[np.isfinite(lv)], 3.25, rtol=1e-6, atol=0) and np.isfinite(lv).sum() >= lv.size - 1
    # Byte conversion
    f = np.concatenate([np.arange(-3, 260, 0.25), [np.nan, np.inf, -np.inf, 1e30, 254.5, 255.49, 0.49999997]]).astype(np.float32)
    for arr in (f.reshape(1, -1), np.arange(-300, 700, dtype=np.int16).reshape(10, 100), np.arange(0, 65536, 257, dtype=np.uint16).reshape(1, -1),
                rng.integers(0, 256, size=shape).astype(np.uint8)):
        assert np.array_equal(D._gdal_byte(arr), co.gdal_byte(arr)), arr.dtype
    # resize
    u = rng.integers(0, 255, size=shape).astype(np.uint8)
    for oh, ow in ((10, 10), shape, (70, 90), (1, 1), (5, 100)):
        assert np.array_equal(geotiff.resample_nearest(u, oh, ow), co.resample_nearest(u, oh, ow)), (oh, ow)
