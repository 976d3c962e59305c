"""SURVEY section 8 f4, the GPU half: the raster formats either side of the per-pixel path on the
device -- COG blocks + NEAREST overviews + predictor (dswx_cog_blocks_device), inflated blocks -> plane
(dswx_untile_device), the Float32 RGB composites (dswx_rgb_planes_device) -- against the host writer / reader's own
whole-array restatements in proteus_amd/geotiff.py (blocked_level, overview_nearest, TiffDirectory.untile), which the
CPU suite pins against files written by another TIFF library.  Bit-exact: integer and byte work."""
import numpy as np
import pytest

from proteus_amd import _capi, geotiff

pytestmark = pytest.mark.gpu

FACTORS = geotiff.COG_OVERVIEW_FACTORS


@pytest.fixture(scope='module')
def ctx():
    c = _capi.Context(0)
    yield c
    c.close()


def _device_blocks(ctx, arr, factors, tile, predictor):
    """arr [H,W] -> the device's blocked buffer (uint8) + the layout."""
    lay = _capi.cog_layout(arr.shape[0], arr.shape[1], arr.dtype.itemsize, factors, tile)
    d_in = ctx.malloc(max(arr.nbytes, 16))
    d_out = ctx.malloc(max(lay['total_bytes'], 16))
    try:
        d_in.upload(arr)
        ctx.lib.dswx_memset_d(ctx.handle, d_out.ptr, 0xAB, d_out.nbytes)           # every byte must be written
        ctx.cog_blocks_device(d_in.ptr, arr.dtype.itemsize, arr.shape[0], arr.shape[1], d_out.ptr, factors, tile, predictor)
        ctx.synchronize()
        return d_out.download(np.uint8, lay['total_bytes']), lay
    finally:
        d_in.free()
        d_out.free()


def _host_levels(arr, factors, tile, predictor):
    levels = [arr] + [geotiff.overview_nearest(arr, f) for f in factors if f > 1 and arr.shape != (1, 1)]
    return [geotiff.blocked_level(lv[None], tile, predictor) for lv in levels]


@pytest.mark.parametrize('shape', [(3660, 3660), (1, 1), (7, 3), (513, 1025), (1000, 333), (129, 4097), (2048, 512)])
@pytest.mark.parametrize('dtype', [np.uint8, np.uint16])
def test_cog_blocks_and_nearest_overviews(ctx, shape, dtype):
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    arr = rng.integers(0, np.iinfo(dtype).max + 1, size=shape).astype(dtype)
    arr[::5] = 3                                                        # runs, as class maps have
    for predictor in (2, 1):
        for tile in ((512, 16) if shape[0] < 2000 else (512,)):
            got, lay = _device_blocks(ctx, arr, FACTORS, tile, predictor)
            want = _host_levels(arr, FACTORS, tile, predictor)
            assert lay['n_levels'] == len(want)
            for k, (lv, host) in enumerate(zip(lay['levels'], want)):
                assert (lv['height'], lv['width']) == (host.height, host.width)
                assert (lv['blocks_down'], lv['blocks_across']) == (host.down, host.across)
                n = host.n_blocks * host.block_bytes
                dev = got[lv['offset_bytes']: lv['offset_bytes'] + n]
                assert np.array_equal(dev, host.data.reshape(-1).view(np.uint8)), (shape, dtype, predictor, tile, k)
            assert lay['total_bytes'] == sum(h.n_blocks * h.block_bytes for h in want)


def test_cog_blocks_without_overviews_and_signed_samples(ctx):
    rng = np.random.default_rng(4)
    arr = rng.integers(-32768, 32768, size=(700, 900)).astype(np.int16)
    got, lay = _device_blocks(ctx, arr.view(np.uint16), (), 512, 2)
    host = geotiff.blocked_level(arr[None], 512, 2)
    assert lay['n_levels'] == 1 and np.array_equal(got, host.data.reshape(-1).view(np.uint8))


@pytest.mark.parametrize('shape', [(3660, 3660), (300, 257), (1, 1), (513, 100)])
def test_float32_blocks_with_the_floating_point_predictor(ctx, shape):
    rng = np.random.default_rng(shape[1])
    arr = rng.normal(0.1, 0.05, size=shape).astype(np.float32)
    arr[rng.random(shape) < 0.05] = np.nan
    arr.reshape(-1)[:3] = [np.inf, -0.0, 1e-40][:arr.size]
    got, lay = _device_blocks(ctx, arr, (), 512, 3)
    host = geotiff.blocked_level(arr[None], 512, 3)
    assert np.array_equal(got, np.asarray(host.data).reshape(-1).view(np.uint8))
    if shape != (1, 1):                 # (a 1 x 1 raster has no overview level whatever the factors)
        with pytest.raises(_capi.DswxError, match='CUBICSPLINE'):
            _device_blocks(ctx, arr, (4,), 512, 3)


def _untile(ctx, staging, dtype, H, W, bw, bh, predictor):
    d_in = ctx.malloc(max(staging.nbytes, 16))
    d_out = ctx.malloc(max(H * W * np.dtype(dtype).itemsize, 16))
    try:
        d_in.upload(staging)
        ctx.lib.dswx_memset_d(ctx.handle, d_out.ptr, 0xCD, d_out.nbytes)
        ctx.untile_device(d_in.ptr, np.dtype(dtype).itemsize, H, W, bw, bh, predictor, d_out.ptr)
        ctx.synchronize()
        return d_out.download(dtype, H * W).reshape(H, W)
    finally:
        d_in.free()
        d_out.free()


@pytest.mark.parametrize('shape', [(3660, 3660), (1, 1), (7, 3), (513, 1025), (1000, 333)])
@pytest.mark.parametrize('dtype', [np.uint8, np.int16])
def test_untile_tiles_and_strips(ctx, tmp_path, shape, dtype):
    """Round trip through REAL files: our writer's tiles (512 and 16... 256) and Pillow / libtiff's DEFLATE strips (short
    last strip), inflated on the host (native codec), untiled + un-predicted on the device == the host reader."""
    rng = np.random.default_rng(shape[0] + 3 * shape[1])
    info = np.iinfo(dtype)
    arr = rng.integers(info.min, info.max + 1, size=shape).astype(dtype)
    H, W = shape
    for tile in (512, 256):
        p = str(tmp_path / f't{tile}.tif')
        geotiff.write_geotiff(p, arr, tile=tile)
        d = geotiff.open_geotiff(p)
        staging = d.inflate()
        got = _untile(ctx, staging, dtype, H, W, d.bw, d.bh, d.predictor)
        assert d.predictor == 2 and np.array_equal(got, arr), (shape, dtype, tile)
        assert np.array_equal(d.untile(staging)[0], arr)
    from PIL import Image, features
    if not features.check('libtiff') or dtype != np.uint8:
        return
    q = str(tmp_path / 'strips.tif')
    Image.fromarray(arr).save(q, compression='tiff_adobe_deflate', tiffinfo={317: 2})
    d = geotiff.open_geotiff(q)
    assert not d.tiled and d.bw == W
    got = _untile(ctx, d.inflate(), dtype, H, W, d.bw, d.bh, d.predictor)
    assert np.array_equal(got, arr)
    # the same raster as an LZW file (what other GDAL tools often hand over): the native codec's LZW decoder, then the device
    Image.fromarray(arr).save(q, compression='tiff_lzw', tiffinfo={317: 2})
    d = geotiff.open_geotiff(q)
    assert d.comp == 5 and d.predictor == 2
    got = _untile(ctx, d.inflate(), dtype, H, W, d.bw, d.bh, d.predictor)
    assert np.array_equal(got, arr)


def test_rgb_planes(ctx):
    """_save_output_rgb_file's arithmetic (dswx_hls.py:3013-3036) in float32, NaN on invalid pixels, clipped bands."""
    rng = np.random.default_rng(12)
    n = 3660 * 37 + 5
    bands = [rng.integers(-200, 12000, size=n).astype(np.int16) for _ in range(3)]
    diag = rng.integers(0, 11112, size=n).astype(np.uint16)
    diag[rng.random(n) < 0.1] = 65535
    for scale, offset, clip in (([1e-4] * 3, [0.0] * 3, True), ([1e-4, 2e-4, 0.5], [0.0, -12.5, 3.0], True),
                                ([1e-4] * 3, [0.0] * 3, False)):
        d = [ctx.malloc(b.nbytes) for b in bands]
        d_diag = ctx.malloc(diag.nbytes)
        d_out = ctx.malloc(3 * n * 4)
        try:
            for buf, b in zip(d, bands):
                buf.upload(b)
            d_diag.upload(diag)
            for use_diag in (True, False):
                ctx.rgb_planes_device(d[0].ptr, d[1].ptr, d[2].ptr, d_diag.ptr if use_diag else None, n, scale, offset, clip,
                                      d_out.ptr)
                ctx.synchronize()
                got = d_out.download(np.float32, 3 * n).reshape(3, n)
                for c in range(3):
                    b = np.clip(bands[c], 1, None) if clip else bands[c]
                    want = scale[c] * (np.asarray(b, dtype=np.float32) - offset[c])       # the reference's statement
                    assert want.dtype == np.float32
                    if use_diag:
                        want[diag == 65535] = np.nan
                    assert np.array_equal(got[c], want, equal_nan=True), (scale, offset, clip, use_diag, c)
        finally:
            for buf in d + [d_diag, d_out]:
                buf.free()


def test_files_from_device_blocks_are_the_host_writers_files(ctx, tmp_path):
    """The whole writer side: a layer resident on the device -> COG through pipeline.TileEngine.layer_levels (device
    blocks + overviews + predictor, host DEFLATE) is BYTE FOR BYTE the file the host writer makes from the same array
    (same blocks -> same DEFLATE streams -> same directory), palette / nodata / metadata included; the same for the
    three-band Float32 composite of _save_output_rgb_file; and the reader side: the engine's resident plane of a band
    file == read_geotiff's array."""
    from proteus_amd import dswx_hls as D
    from proteus_amd import pipeline
    eng = pipeline.TileEngine(ctx)
    rng = np.random.default_rng(31)
    geo = geotiff.geo_tags_from_geotransform((600000.0, 30.0, 0.0, 4000020.0, 0.0, -30.0), epsg=32615)
    md = {'PRODUCT_ID': 'x', 'N': 5}
    try:
        for shape, dtype in (((3660, 3660), np.uint8), ((1000, 333), np.uint8), ((517, 1031), np.uint16), ((1, 1), np.uint8)):
            arr = rng.integers(0, 5, size=shape).astype(dtype) * (1 if dtype == np.uint8 else 1111)
            ct = {0: (255, 255, 255), 1: (0, 0, 255), 255: (0, 0, 0)} if dtype == np.uint8 else None
            p_host, p_dev = str(tmp_path / 'host.tif'), str(tmp_path / 'dev.tif')
            geotiff.write_geotiff(p_host, arr, geo_tags=geo, metadata=md, nodata=255, descriptions=['layer'], colormap=ct,
                                  overviews=FACTORS)
            plane = eng.upload(arr)
            geotiff.write_geotiff(p_dev, None, levels=eng.layer_levels(plane, FACTORS), geo_tags=geo, metadata=md, nodata=255,
                                  descriptions=['layer'], colormap=ct)
            assert open(p_host, 'rb').read() == open(p_dev, 'rb').read(), (shape, dtype)
            assert geotiff.validate_cog(p_dev) == []
            # reader side: the file back into a resident plane
            back, info = eng.read_plane(p_dev)
            assert back.shape == shape and back.dtype == dtype and np.array_equal(back.numpy(), arr)
            assert info.nodata == 255.0 and info.metadata['PRODUCT_ID'] == 'x'
        # Float32 layer (the DEM layer): CUBICSPLINE overviews, floating-point predictor, NaN nodata
        for shape in ((3660, 3660), (700, 333), (1, 1)):
            dem = (rng.normal(size=shape) * 300 + 500).astype(np.float32)
            dem[rng.random(shape) < 0.02] = np.nan
            p_host, p_dev = str(tmp_path / 'dem_host.tif'), str(tmp_path / 'dem_dev.tif')
            geotiff.write_geotiff(p_host, dem, geo_tags=geo, metadata=md, nodata=float('nan'), descriptions=['dem'],
                                  overviews=FACTORS)
            geotiff.write_geotiff(p_dev, None, levels=eng.layer_levels(eng.upload(dem), FACTORS), geo_tags=geo, metadata=md,
                                  nodata=float('nan'), descriptions=['dem'])
            assert open(p_host, 'rb').read() == open(p_dev, 'rb').read(), shape
            assert geotiff.validate_cog(p_dev) == []
            if shape != (1, 1):
                ovr, _ = geotiff.read_geotiff(p_dev, overview=len(FACTORS) - 1)
                assert ovr.shape == tuple(-(-n // FACTORS[-1]) for n in shape)
        # RGB composite: host statement vs device planes, files identical (three bands, CUBICSPLINE overviews of each)
        h, w = 700, 900
        bands = {k: rng.integers(-50, 9000, size=(h, w)).astype(np.int16) for k in ('red', 'green', 'blue')}
        diag = rng.integers(0, 11112, size=(h, w)).astype(np.uint16)
        diag[rng.random((h, w)) < 0.05] = 65535
        scale = {'red': 1e-4, 'green': 2e-4, 'blue': 1e-4}
        offset = {'red': 0.0, 'green': -3.0, 'blue': 12.5}
        p_host, p_dev = str(tmp_path / 'rgb_host.tif'), str(tmp_path / 'rgb_dev.tif')
        clipped = {k: np.clip(v, 1, None) for k, v in bands.items()}
        D._save_output_rgb_file(clipped['red'], clipped['green'], clipped['blue'], p_host, offset, scale, False, md, geo,
                                invalid_mask=diag == 65535)
        planes = [eng.upload(bands[k]) for k in ('red', 'green', 'blue')]
        D._save_output_rgb_planes(eng, planes, eng.upload(diag), [scale[k] for k in ('red', 'green', 'blue')],
                                  [offset[k] for k in ('red', 'green', 'blue')], p_dev, md, geo)
        assert open(p_host, 'rb').read() == open(p_dev, 'rb').read()
        rgb, _ = geotiff.read_geotiff(p_dev)
        assert rgb.shape == (3, h, w) and np.isnan(rgb[:, diag == 65535]).all()
        assert geotiff.validate_cog(p_dev) == []
        small, _ = geotiff.read_geotiff(p_dev, overview=0)
        assert small.shape == (3, -(-h // 4), -(-w // 4)) and np.isfinite(small).any()
    finally:
        eng.close()


def test_writer_kernels_fuzz(ctx):
    """300 random geometries: raster sizes 1 ... 1400, tiles 8 ... 512, factor sets with odd and repeated factors and the
    reference's, both predictors, u8 / u16 -- dswx_cog_blocks_device against blocked_level(overview_nearest(...)); and the
    inverse on random BLOCK shapes (tiles whose width is not a multiple of 8, strips, blocks larger than the raster):
    dswx_untile_device of blocked_level's own output gives the raster back."""
    rng = np.random.default_rng(20260606)
    for case in range(300):
        h, w = (int(rng.integers(1, 1400)), int(rng.integers(1, 1400))) if case % 5 else (int(rng.integers(1, 40)), int(rng.integers(1, 40)))
        dtype = (np.uint8, np.uint16)[case % 2]
        tile = int(rng.choice([8, 16, 24, 64, 256, 512]))
        factors = [tuple(FACTORS), (2,), (3, 5, 7), (4, 4), (128, 2), ()][int(rng.integers(0, 6))]
        predictor = 1 + case % 2 if case % 7 else 2
        arr = rng.integers(0, np.iinfo(dtype).max + 1, size=(h, w)).astype(dtype)
        if case % 3 == 0:
            arr = (arr % 5).astype(dtype)
        got, lay = _device_blocks(ctx, arr, factors, tile, predictor)
        want = _host_levels(arr, factors, tile, predictor)
        assert lay['n_levels'] == len(want), (case, h, w, factors)
        for k, (lv, host) in enumerate(zip(lay['levels'], want)):
            n = host.n_blocks * host.block_bytes
            assert np.array_equal(got[lv['offset_bytes']: lv['offset_bytes'] + n], host.data.reshape(-1).view(np.uint8)), \
                (case, h, w, dtype, tile, factors, predictor, k)
        # the inverse, on a block shape of its own (bw x bh: any width, strips when bw >= w)
        bw = int(rng.choice([w, int(rng.integers(1, 700)), 8 * int(rng.integers(1, 80))]))
        bh = int(rng.integers(1, 300))
        across, down = -(-w // bw), -(-h // bh)
        pad = np.zeros((down * bh, across * bw), dtype)
        pad[:h, :w] = arr
        blk = np.ascontiguousarray(pad.reshape(down, bh, across, bw).transpose(0, 2, 1, 3))
        if predictor == 2:
            d = blk.copy()
            d[..., 1:] -= blk[..., :-1]
            blk = d
        back = _untile(ctx, blk.reshape(-1), dtype, h, w, bw, bh, predictor)
        assert np.array_equal(back, arr), (case, h, w, dtype, bw, bh, predictor)


@pytest.mark.parametrize('shape', [(3760, 3760), (300, 257), (1, 1), (513, 100)])
def test_untile_float32_with_the_floating_point_predictor_and_four_byte_integers(ctx, tmp_path, shape):
    """A Float32 DEM as GDAL writes it (PREDICTOR=3: byte planes, MSB first, byte-wise running sum; TIFF Technical Note 3)
    read back on the device: our writer's tiles and Pillow / libtiff's strips; bit for bit incl. NaN / inf / denormals.
    And 4-byte integer samples with PREDICTOR=2 (wrap-around in 32 bits).  Then the crop of a margin on the device."""
    from proteus_amd import pipeline
    rng = np.random.default_rng(shape[0] + shape[1])
    arr = rng.normal(300.0, 120.0, size=shape).astype(np.float32)
    arr[rng.random(shape) < 0.02] = np.nan
    arr.reshape(-1)[:3] = [np.inf, -0.0, 1e-40][:arr.size]
    eng = pipeline.TileEngine(ctx)
    try:
        for tile in (512, 256):
            p = str(tmp_path / f'f{tile}.tif')
            geotiff.write_geotiff(p, arr, tile=tile, nodata=float('nan'))
            d = geotiff.open_geotiff(p)
            assert d.predictor == 3 and eng.device_untile_ok(d)
            plane, info = eng.read_directory(d)
            assert plane.dtype == np.float32 and plane.numpy().tobytes() == arr.tobytes(), (shape, tile)
        from PIL import Image, features
        if features.check('libtiff'):
            q = str(tmp_path / 'strips.tif')
            Image.fromarray(arr).save(q, compression='tiff_adobe_deflate', tiffinfo={317: 3})
            d = geotiff.open_geotiff(q)
            assert d.predictor == 3 and not d.tiled and eng.device_untile_ok(d)
            assert eng.read_directory(d)[0].numpy().tobytes() == arr.tobytes()
        ints = rng.integers(-2 ** 31, 2 ** 31, size=shape).astype(np.int32)
        p = str(tmp_path / 'i32.tif')
        geotiff.write_geotiff(p, ints)
        d = geotiff.open_geotiff(p)
        assert d.predictor == 2 and d.dt.itemsize == 4 and eng.device_untile_ok(d)
        assert np.array_equal(eng.read_directory(d)[0].numpy(), ints)
        if min(shape) > 8:
            m = 3
            plane = eng.upload(arr)
            assert eng.crop(plane, m).numpy().tobytes() == np.ascontiguousarray(arr[m:-m, m:-m]).tobytes()
            assert eng.crop(plane, 0) is plane
    finally:
        eng.close()


@pytest.mark.parametrize('shape', [(1, 1), (7, 3), (65, 130), (150, 201), (259, 77)])
def test_device_kernels_against_the_row_by_row_oracle(ctx, shape):
    """The device kernels against oracle/cog_oracle.py -- the block-by-block, row-by-row restatement that shares no code
    with the product's writer and is pinned both ways against libtiff (tests/test_cog_oracle.py): block bytes of every
    level (u8 / u16 with PREDICTOR 2 and 1, Float32 with PREDICTOR 3), the way back (tiles and strips, incl. Float32), and
    the RGB planes."""
    from oracle import cog_oracle as co
    rng = np.random.default_rng(shape[0] * 13 + shape[1])
    for dtype in (np.uint8, np.uint16):
        arr = rng.integers(0, np.iinfo(dtype).max + 1, size=shape).astype(dtype)
        for tile, predictor in ((16, 2), (64, 2), (16, 1)):
            got, lay = _device_blocks(ctx, arr, FACTORS, tile, predictor)
            want = co.cog_levels(arr, FACTORS, tile, predictor)
            assert lay['n_levels'] == len(want)
            for lv, (h, w, data) in zip(lay['levels'], want):
                assert (lv['height'], lv['width']) == (h, w)
                assert np.array_equal(got[lv['offset_bytes']: lv['offset_bytes'] + data.size], data), (dtype, tile, predictor)
            back = _untile(ctx, want[0][2], dtype, shape[0], shape[1], tile, tile, predictor)
            assert np.array_equal(back, arr)
        # strips: one block row of the raster's own width, several rows per strip, the last strip short
        bh = 7
        down = -(-shape[0] // bh)
        pad = np.zeros((down * bh, shape[1]), dtype)
        pad[:shape[0]] = arr
        d = pad.copy()
        d[:, 1:] -= pad[:, :-1]
        assert np.array_equal(co.unblocks(d.reshape(-1).view(np.uint8), dtype, shape[0], shape[1], shape[1], bh, 2), arr)
        assert np.array_equal(_untile(ctx, d.reshape(-1), dtype, shape[0], shape[1], shape[1], bh, 2), arr)
    f = rng.normal(10.0, 4.0, size=shape).astype(np.float32)
    f[rng.random(shape) < 0.1] = np.nan
    got, lay = _device_blocks(ctx, f, (), 16, 3)
    (h, w, data), = co.cog_levels(f, (), 16, 3)
    assert np.array_equal(got, data)
    d_in, d_out = ctx.malloc(max(data.size, 16)), ctx.malloc(max(f.nbytes, 16))
    try:
        d_in.upload(data)
        ctx.untile_device(d_in.ptr, 4, shape[0], shape[1], 16, 16, 3, d_out.ptr)
        ctx.synchronize()
        assert d_out.download(np.float32, f.size).tobytes() == f.tobytes()
    finally:
        d_in.free()
        d_out.free()
    # RGB planes
    n = shape[0] * shape[1]
    bands = [rng.integers(-100, 9000, size=n).astype(np.int16) for _ in range(3)]
    diag = rng.integers(0, 11112, size=n).astype(np.uint16)
    diag[::5] = 65535
    bufs = [ctx.malloc(max(b.nbytes, 16)) for b in bands] + [ctx.malloc(max(diag.nbytes, 16)), ctx.malloc(max(12 * n, 16))]
    try:
        for buf, b in zip(bufs, bands + [diag]):
            buf.upload(b)
        ctx.rgb_planes_device(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, n, [1e-4, 2e-4, 1.0], [0.0, -3.0, 12.5], True, bufs[4].ptr)
        ctx.synchronize()
        want = co.rgb_planes(bands, diag, [1e-4, 2e-4, 1.0], [0.0, -3.0, 12.5])
        assert np.array_equal(bufs[4].download(np.float32, 3 * n).reshape(3, n), want, equal_nan=True)
    finally:
        for buf in bufs:
            buf.free()


@pytest.mark.parametrize('shape', [(3660, 3660), (1000, 333), (7, 3), (513, 1025), (4, 4), (1, 5), (129, 4097)])
def test_cubicspline_overview_pyramid_on_the_device(ctx, shape):
    """save_as_cog's overviews of a non-integer layer (core.py:41-46: CUBICSPLINE, 4 / 16 / 64 / 128, cascaded) from a
    plane resident on the device (dswx_convolve_axis_device, horizontal pass into float64, vertical pass, float32) against
    the host writer's statement (geotiff.overview_cubicspline, the same weights, the taps in the same order): every level
    bit for bit, NaN where the whole support is NaN, infinities kept."""
    from proteus_amd import pipeline
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    a = (rng.normal(size=shape) * 1000).astype(np.float32)
    a[rng.random(shape) < 0.03] = np.nan
    if shape[0] > 200:
        a[40:140, 60:190] = np.nan                      # a hole larger than the support of every level-1 pixel inside it
        a[150, 3] = np.inf
        a[160, 9] = -np.inf
    eng = pipeline.TileEngine(ctx)
    try:
        got = eng._float_pyramid(eng.upload(a), FACTORS)
        want, prev_f = [a], 1
        for f in FACTORS:                               # write_geotiff's cascade
            lv = geotiff.overview_cubicspline(want[-1], f // prev_f) if prev_f > 1 and f % prev_f == 0 \
                else geotiff.overview_cubicspline(a, f)
            if lv.shape != tuple(-(-n // f) for n in shape):
                lv = geotiff.overview_cubicspline(a, f)
            want.append(lv)
            prev_f = f
        if shape == (1, 1):
            want = want[:1]
        assert len(got) == len(want)
        for k, (g, w) in enumerate(zip(got, want)):
            g = g.numpy()
            assert g.shape == w.shape and g.dtype == np.float32 == w.dtype, (k, g.shape, w.shape)
            assert np.array_equal(np.isnan(g), np.isnan(w)), k
            assert np.array_equal(g, w, equal_nan=True), (k, np.nanmax(np.abs(g.astype(np.float64) - w)))
        if shape[0] > 200:
            assert np.isnan(got[1].numpy()).any() and np.isinf(got[1].numpy()).any()
    finally:
        eng.close()


def test_convolve_axis_refuses_bad_arguments(ctx):
    d = ctx.malloc(64)
    try:
        with pytest.raises(_capi.DswxError):
            ctx.convolve_axis_device(d.ptr, False, 1, 0, 1, 1, 1, 1, d.ptr, d.ptr, d.ptr, False, 1, 1)        # n_in < 1
        with pytest.raises(_capi.DswxError):
            ctx.convolve_axis_device(d.ptr, False, 1, 4, 4, 1, 1, 0, d.ptr, d.ptr, d.ptr, False, 1, 1)        # no taps
        with pytest.raises(_capi.DswxError):
            ctx.convolve_axis_device(None, False, 1, 4, 4, 1, 1, 1, d.ptr, d.ptr, d.ptr, False, 1, 1)
    finally:
        d.free()


def test_multiband_product_from_resident_layers_is_the_host_writers_file(ctx, tmp_path):
    """save_dswx_product (dswx_hls.py:2601-2707): ten Byte bands, DIAG and DEM through GDAL's Byte conversion, absent / None
    layers, descriptions -- from planes resident on the device (dswx_to_byte_device + pipeline.band_stack_levels) the file is
    byte for byte the one the host mirror writes from the same arrays; the conversion itself against dswx_hls._gdal_byte."""
    from proteus_amd import dswx_hls as D
    from proteus_amd import pipeline
    eng = pipeline.TileEngine(ctx)
    rng = np.random.default_rng(77)
    geo = geotiff.geo_tags_from_geotransform((600000.0, 30.0, 0.0, 4000020.0, 0.0, -30.0), epsg=32615)
    md = {'PRODUCT_ID': 'mb'}
    try:
        # the conversion: every uint16 / int16 value; floats around every rounding edge, NaN, infinities
        u16 = np.arange(65536, dtype=np.uint16).reshape(256, 256)
        i16 = u16.view(np.int16)
        f32 = np.concatenate([np.arange(-4, 260, 0.25, dtype=np.float32), np.float32([np.nan, np.inf, -np.inf, 254.5, 255.49, 255.5, 1e30, -1e30,
                                                                                      0.49999997, 0.5, 1.5, 2.5, 254.49998]),
                              (rng.normal(size=4000) * 200).astype(np.float32)])
        f32 = np.resize(f32, (72, 80)).astype(np.float32)
        for a in (u16, i16, f32):
            got = eng.byte_plane(eng.upload(a)).numpy()
            assert got.dtype == np.uint8 and np.array_equal(got, D._gdal_byte(a)), a.dtype
        for shape in ((1100, 700), (3660, 3660)):
            u8 = lambda hi: rng.integers(0, hi, size=shape).astype(np.uint8)          # noqa: E731
            dem = (rng.normal(size=shape) * 150 + 100).astype(np.float32)
            dem[rng.random(shape) < 0.01] = np.nan
            layers = {'WTR': u8(3), 'BWTR': u8(2), 'DIAG': rng.integers(0, 11112, size=shape).astype(np.uint16), 'WTR-1': u8(5),
                      'WTR-2': u8(5), 'LAND': None, 'SHAD': u8(2), 'CLOUD': u8(8), 'DEM': dem}
            for drop in ((), ('LAND', 'DEM')):
                use = {k: v for k, v in layers.items() if k not in drop}
                p_host, p_dev = str(tmp_path / 'mb_host.tif'), str(tmp_path / 'mb_dev.tif')
                D.save_dswx_product(use, p_host, md, geo)
                D._save_dswx_product_planes(eng, {k: (None if v is None else eng.upload(v)) for k, v in use.items()}, p_dev, md, geo)
                assert open(p_host, 'rb').read() == open(p_dev, 'rb').read(), (shape, drop)
                assert geotiff.validate_cog(p_dev) == []
            arr, info = geotiff.read_geotiff(p_dev)
            assert arr.shape == (10,) + shape and np.array_equal(arr[2], np.minimum(layers['DIAG'], 255))
        with pytest.raises(ValueError):
            D._save_dswx_product_planes(eng, {'WTR': None}, str(tmp_path / 'x.tif'), md, geo)
    finally:
        eng.close()


@pytest.mark.parametrize('dtype', [np.uint8, np.uint16, np.float32])
def test_browse_resampling_in_hbm(ctx, dtype):
    """geotiff2png's nearest-neighbour resize (GDAL RasterIO's pick, geotiff.resample_nearest) as a gather on the resident
    plane: decimation, identity, magnification, one pixel."""
    from proteus_amd import pipeline
    eng = pipeline.TileEngine(ctx)
    rng = np.random.default_rng(5)
    try:
        for shape, outs in (((3660, 3660), ((1024, 1024), (3660, 3660), (1, 1), (333, 2048))), ((7, 5), ((14, 9), (3, 5), (1, 2)))):
            a = rng.integers(0, 250, size=shape).astype(dtype)
            plane = eng.upload(a)
            for oh, ow in outs:
                got = eng.resample_nearest(plane, oh, ow)
                assert got.dtype == a.dtype and np.array_equal(got, geotiff.resample_nearest(a, oh, ow)), (shape, oh, ow)
    finally:
        eng.close()


@pytest.mark.parametrize('shape', [(37, 53), (5, 3), (1, 7), (130, 90), (259, 77)])
def test_round6_writer_kernels_against_the_element_by_element_oracle(ctx, shape):
    """dswx_convolve_axis_device (the CUBICSPLINE pyramid), dswx_to_byte_device, dswx_gather_2d_device against
    oracle/cog_oracle.py's restatements of the three GDAL rules (element by element, no code shared with the product)."""
    from oracle import cog_oracle as co
    from proteus_amd import pipeline
    rng = np.random.default_rng(shape[0] * 17 + shape[1])
    a = (rng.normal(size=shape) * 1000).astype(np.float32)
    a[rng.random(shape) < 0.05] = np.nan
    if shape[0] > 30:
        a[3:22, 4:30] = np.nan
        a[25, 7], a[27, 9] = np.inf, -np.inf
    eng = pipeline.TileEngine(ctx)
    try:
        got = eng._float_pyramid(eng.upload(a), FACTORS)
        want = co.cubicspline_pyramid(a, FACTORS)
        assert len(got) == len(want)
        for k, (g, w) in enumerate(zip(got, want)):
            assert np.array_equal(g.numpy(), w, equal_nan=True), k
        f = np.resize(np.concatenate([np.arange(-3, 260, 0.25), [np.nan, np.inf, -np.inf, 1e30, 254.5, 255.49]]).astype(np.float32), shape)
        for arr in (f, rng.integers(-300, 700, size=shape).astype(np.int16), rng.integers(0, 65536, size=shape).astype(np.uint16)):
            assert np.array_equal(eng.byte_plane(eng.upload(arr)).numpy(), co.gdal_byte(arr)), arr.dtype
        u = rng.integers(0, 255, size=shape).astype(np.uint8)
        plane = eng.upload(u)
        for oh, ow in ((10, 10), shape, (70, 90), (1, 1), (5, 100)):
            assert np.array_equal(eng.resample_nearest(plane, oh, ow), co.resample_nearest(u, oh, ow)), (oh, ow)
    finally:
        eng.close()


def test_cubicspline_weight_cache_starts_over_without_losing_a_launch(ctx):
    """The device copies of the convolution weights are cached per (n_in, n_out) and the cache is emptied when it outgrows
    its cap: with a cap of four pairs, pyramids of many different sizes in a row -- each level still bit for bit the host's."""
    from proteus_amd import pipeline
    eng = pipeline.TileEngine(ctx)
    eng.WEIGHTS_CAP = 4
    rng = np.random.default_rng(99)
    try:
        for k in range(12):
            shape = (int(rng.integers(5, 90)), int(rng.integers(5, 90)))
            a = rng.normal(size=shape).astype(np.float32)
            got = eng._float_pyramid(eng.upload(a), FACTORS)
            want, prev = [a], 1
            for f in FACTORS:
                lv = geotiff.overview_cubicspline(want[-1], f // prev) if prev > 1 and f % prev == 0 else geotiff.overview_cubicspline(a, f)
                if lv.shape != tuple(-(-n // f) for n in shape):
                    lv = geotiff.overview_cubicspline(a, f)
                want.append(lv)
                prev = f
            assert len(eng._weights) <= 4
            for g, w in zip(got, want):
                assert np.array_equal(g.numpy(), w, equal_nan=True), (k, shape)
    finally:
        eng.close()
