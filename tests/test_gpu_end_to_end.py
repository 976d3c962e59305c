"""BASELINE.json configs[0]/[1] plumbing on the GPU: synthetic HLS GeoTIFFs + runconfig
-> bin/dswx_hls.py -> GeoTIFF layers, checked against the oracle; plus the reference's own
unit test (tests/test_dswx_hls_units.py) restated against this package."""
import os
import subprocess
import time
import sys

import numpy as np
import pytest

from oracle import dswx_oracle as o
from proteus_amd import dswx_hls as D
from proteus_amd import _capi, geotiff

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_synthetic_hls as synth_hls   # noqa: E402

LAYER_FILES = {'WTR': 'B01_WTR', 'BWTR': 'B02_BWTR', 'CONF': 'B03_CONF', 'DIAG': 'B04_DIAG',
               'WTR-1': 'B05_WTR-1', 'WTR-2': 'B06_WTR-2', 'CLOUD': 'B09_CLOUD'}


def test_units():
    """Same body as the reference's tests/test_dswx_hls_units.py:7-28."""
    from proteus_amd.dswx_hls import interpreted_dswx_band_dict, generate_interpreted_layer
    length = 1
    width = len(interpreted_dswx_band_dict) + 1
    input_array = np.full((length, width), 111111)
    expected_output_array = np.full((length, width), 255)
    for i, (key, value) in enumerate(interpreted_dswx_band_dict.items()):
        input_array[0, i] = key
        expected_output_array[0, i] = value
    output_array = generate_interpreted_layer(input_array)
    assert np.array_equal(output_array, expected_output_array)


def test_workflow(tmp_path):
    """The reference's tests/test_dswx_hls_workflow.py:17-120 restated: same parser, same
    create_logger / parse_runconfig_file / generate_dswx_layers call with the full keyword
    list, same compare_dswx_hls_products loop over a ref_dir.  The Zenodo dataset is replaced
    by a synthetic S30 tile and the ref_dir is written from the oracle's layers."""
    from proteus_amd.dswx_hls import (get_dswx_hls_cli_parser, generate_dswx_layers, create_logger,
                                      parse_runconfig_file, compare_dswx_hls_products)
    dataset_dir = str(tmp_path / 's30_synthetic')
    user_runconfig_file, _, _, s = synth_hls.make(dataset_dir, sensor='S30', size=1100, tile=31)
    output_dir = os.path.join(dataset_dir, 'output')
    ref_dir = os.path.join(dataset_dir, 'ref_dir')

    parser = get_dswx_hls_cli_parser()
    args = parser.parse_args([user_runconfig_file])
    create_logger(args.log_file)
    runconfig_constants = parse_runconfig_file(user_runconfig_file=user_runconfig_file, args=args)
    args.flag_debug = True            # as the reference test: the 1000 x 1000 window

    assert generate_dswx_layers(
        args.input_list,
        args.output_file,
        hls_thresholds=runconfig_constants.hls_thresholds,
        dem_file=args.dem_file,
        dem_file_description=args.dem_file_description,
        output_interpreted_band=args.output_interpreted_band,
        output_rgb_file=args.output_rgb_file,
        output_infrared_rgb_file=args.output_infrared_rgb_file,
        output_binary_water=args.output_binary_water,
        output_confidence_layer=args.output_confidence_layer,
        output_diagnostic_layer=args.output_diagnostic_layer,
        output_non_masked_dswx=args.output_non_masked_dswx,
        output_shadow_masked_dswx=args.output_shadow_masked_dswx,
        output_landcover=args.output_landcover,
        output_shadow_layer=args.output_shadow_layer,
        output_cloud_layer=args.output_cloud_layer,
        output_dem_layer=args.output_dem_layer,
        output_browse_image=args.output_browse_image,
        browse_image_height=args.browse_image_height,
        browse_image_width=args.browse_image_width,
        landcover_file=args.landcover_file,
        landcover_file_description=args.landcover_file_description,
        worldcover_file=args.worldcover_file,
        worldcover_file_description=args.worldcover_file_description,
        shoreline_shapefile=args.shoreline_shapefile,
        shoreline_shapefile_description=args.shoreline_shapefile_description,
        flag_offset_and_scale_inputs=args.flag_offset_and_scale_inputs,
        scratch_dir=args.scratch_dir,
        product_id=args.product_id,
        product_version=args.product_version,
        check_ancillary_inputs_coverage=args.check_ancillary_inputs_coverage,
        apply_aerosol_class_remapping=args.apply_aerosol_class_remapping,
        aerosol_not_water_to_high_conf_water_fmask_values=
            args.aerosol_not_water_to_high_conf_water_fmask_values,
        aerosol_water_moderate_conf_to_high_conf_water_fmask_values=
            args.aerosol_water_moderate_conf_to_high_conf_water_fmask_values,
        aerosol_partial_surface_water_conservative_to_high_conf_water_fmask_values=
            args.aerosol_partial_surface_water_conservative_to_high_conf_water_fmask_values,
        aerosol_partial_surface_aggressive_to_high_conf_water_fmask_values=
            args.aerosol_partial_surface_aggressive_to_high_conf_water_fmask_values,
        shadow_masking_algorithm=args.shadow_masking_algorithm,
        min_slope_angle=args.min_slope_angle,
        max_sun_local_inc_angle=args.max_sun_local_inc_angle,
        mask_adjacent_to_cloud_mode=args.mask_adjacent_to_cloud_mode,
        forest_mask_landcover_classes=args.forest_mask_landcover_classes,
        ocean_masking_shoreline_distance_km=args.ocean_masking_shoreline_distance_km,
        flag_debug=args.flag_debug)

    # ref_dir: the oracle's layers on the same 1000 x 1000 window, written with the product's
    # own metadata (the comparison ignores only the keys the reference's comparison ignores)
    exp = o.classify_tile([b[:1000, :1000] for b in s['bands']], s['fmask'][:1000, :1000], collapse=True)
    os.makedirs(ref_dir)
    n_ref = 0
    for layer, stem in LAYER_FILES.items():
        name = f'dswx_hls_synth_v1.0_{stem}.tif'
        _, info = geotiff.read_geotiff(os.path.join(output_dir, name))
        geotiff.write_geotiff(os.path.join(ref_dir, name), exp[layer], geo_tags=info.geo_tags,
                              metadata=info.metadata, nodata=info.nodata, descriptions=info.descriptions)
        n_ref += 1
    import glob
    ref_files = glob.glob(os.path.join(ref_dir, '*'))
    assert len(ref_files) == n_ref == 7
    for ref_file in ref_files:
        ref_basename = os.path.basename(ref_file)
        output_file = os.path.join(output_dir, ref_basename)
        assert compare_dswx_hls_products(ref_file, output_file)
    # and the comparison does fail on a one-pixel difference
    arr, info = geotiff.read_geotiff(ref_files[0])
    arr = arr.copy()
    arr[10, 10] ^= 1
    geotiff.write_geotiff(ref_files[0], arr, geo_tags=info.geo_tags, metadata=info.metadata,
                          nodata=info.nodata, descriptions=info.descriptions)
    assert not compare_dswx_hls_products(ref_files[0], os.path.join(output_dir, os.path.basename(ref_files[0])))


@pytest.mark.parametrize('sensor', ['L30', 'S30'])
def test_runconfig_entry_point(tmp_path, sensor):
    rcfile, files, _, s = synth_hls.make(str(tmp_path), sensor=sensor, size=512, tile=9)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'dswx_hls.py'), rcfile,
                          '--log', str(tmp_path / 'run.log')],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    exp = o.classify_tile(s['bands'], s['fmask'], collapse=True)
    out_dir = tmp_path / 'output'
    for layer, stem in LAYER_FILES.items():
        path = out_dir / f'dswx_hls_synth_v1.0_{stem}.tif'
        assert path.exists(), (layer, sorted(os.listdir(out_dir)))
        arr, info = geotiff.read_geotiff(str(path))
        assert np.array_equal(arr, exp[layer]), layer
        assert arr.dtype == (np.uint16 if layer == 'DIAG' else np.uint8)
        assert info.nodata == (65535 if layer == 'DIAG' else 255)
        assert info.descriptions == [D.band_description_dict[layer]]
        assert info.geotransform == (600000.0, 30.0, 0.0, 4000020.0, 0.0, -30.0)
        # save_as_cog (core.py:7-91): cloud-optimized layout with NEAREST overviews
        assert geotiff.validate_cog(str(path)) == []
        assert [x['width'] for x in geotiff.cog_layout(str(path))] == [512, 128, 32, 8, 4]
        ovr, _ = geotiff.read_geotiff(str(path), overview=0)
        assert np.array_equal(ovr, exp[layer][::4, ::4])
        md = info.metadata
        c = exp['counters']
        assert md['SPATIAL_COVERAGE'] == str(c['SPATIAL_COVERAGE'])
        assert md['CLOUD_COVERAGE'] == str(c['CLOUD_COVERAGE'])
        assert md['SPATIAL_COVERAGE_EXCLUDING_MASKED_OCEAN'] == \
            str(c['SPATIAL_COVERAGE_EXCLUDING_MASKED_OCEAN'])
        assert md['PRODUCT_ID'] == 'dswx_hls_synth' and md['PRODUCT_VERSION'] == '1.0'
        assert md['SPACECRAFT_NAME'] == ('Landsat-8' if sensor == 'L30' else 'Sentinel-2A')
        assert md['SENSOR'] == ('OLI' if sensor == 'L30' else 'MSI')
        assert md['HLS_DATASET'] == f'HLS.{sensor}.T15SYU.2021250T163901.v2.0'
        assert md['MASK_ADJACENT_TO_CLOUD_MODE'] == 'mask'
        assert md['INPUT_HLS_PRODUCT_CLOUD_COVERAGE'] == '20'
    assert not (out_dir / 'dswx_hls_synth_v1.0_B07_LAND.tif').exists()
    assert 'per-pixel chain on GPU: dswx_classify' in (tmp_path / 'run.log').read_text()
    wtr = str(out_dir / 'dswx_hls_synth_v1.0_B01_WTR.tif')
    assert D.compare_dswx_hls_products(wtr, wtr)


def test_runconfig_entry_point_at_full_tile_size(tmp_path):
    """BASELINE.json configs[0] at ITS size: one synthetic 3660 x 3660 HLS.L30 tile as seven DEFLATE
    GeoTIFFs through `bin/dswx_hls.py <runconfig>` -- a child process, as a user runs it -- every product layer read back
    and compared, whole, with the numpy oracle's layers of the same tile; coverage metadata from the counters; COG layout
    of a 3660-wide layer."""
    rcfile, files, _, s = synth_hls.make(str(tmp_path), sensor='L30', size=3660, tile=21)
    t0 = time.perf_counter()
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'dswx_hls.py'), rcfile,
                          '--log', str(tmp_path / 'run.log')],
                         capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    exp = o.classify_tile(s['bands'], s['fmask'], collapse=True)
    out_dir = tmp_path / 'output'
    for layer, stem in LAYER_FILES.items():
        path = out_dir / f'dswx_hls_synth_v1.0_{stem}.tif'
        assert path.exists(), (layer, sorted(os.listdir(out_dir)))
        arr, info = geotiff.read_geotiff(str(path))
        assert arr.shape == (3660, 3660) and np.array_equal(arr, exp[layer]), layer
        assert geotiff.validate_cog(str(path)) == []
        c = exp['counters']
        assert info.metadata['SPATIAL_COVERAGE'] == str(c['SPATIAL_COVERAGE'])
        assert info.metadata['CLOUD_COVERAGE'] == str(c['CLOUD_COVERAGE'])
    assert [x['width'] for x in geotiff.cog_layout(str(out_dir / 'dswx_hls_synth_v1.0_B01_WTR.tif'))][:2] == [3660, 915]
    log = (tmp_path / 'run.log').read_text()
    assert 'per-pixel chain on GPU: dswx_classify_lut' in log
    print(f'bin/dswx_hls.py on a 3660 x 3660 tile: {wall:.1f} s wall (child process, incl. interpreter start and GeoTIFF codec)')


def test_api_with_masks_and_multiband(tmp_path):
    rcfile, files, masks, s = synth_hls.make(str(tmp_path), size=300, tile=4, masks=True)
    out = str(tmp_path / 'product.tif')
    ok = D.generate_dswx_layers(
        files, out, apply_ocean_masking=True, mask_adjacent_to_cloud_mode='ignore',
        landcover_mask=masks['land'], shadow_layer=s['shad'].astype(bool),
        ocean_mask=masks['ocean'], output_confidence_layer=str(tmp_path / 'conf.tif'),
        output_shadow_layer=str(tmp_path / 'shad.tif'), output_landcover=str(tmp_path / 'land.tif'))
    assert ok is True
    exp = o.classify_tile(s['bands'], s['fmask'], landcover=s['land'], shadow=s['shad'],
                          ocean_mask=s['ocean'], mask_adjacent_to_cloud_mode='ignore')
    conf, _ = geotiff.read_geotiff(str(tmp_path / 'conf.tif'))
    assert np.array_equal(conf, exp['CONF'])
    stack, info = geotiff.read_geotiff(out)
    # the reference's multi-band file: ten Byte bands created (:2663-2666), nine written -- the call never passes
    # conf and the loop skips it without advancing the band index (:5383-5397, :2673-2686); WTR-1 is the
    # post-aerosol one; DIAG saturates like GDT_Byte; no DEM here -> nodata plane; band 10 stays unwritten;
    # every written band carries the first band's description (never reset, :2686-2687)
    written = ['WTR', 'BWTR', 'DIAG', 'WTR-1', 'WTR-2', 'LAND', 'SHAD', 'CLOUD', 'DEM']
    assert info.bands == 10 and info.descriptions == [D.band_description_dict['WTR']] * 9 + ['']
    want = {'WTR': exp['WTR'], 'BWTR': exp['BWTR'],
            'DIAG': np.minimum(exp['DIAG'], 255).astype(np.uint8),
            'WTR-1': exp['WTR-1-AEROSOL'], 'WTR-2': exp['WTR-2'], 'LAND': s['land'],
            'SHAD': s['shad'], 'CLOUD': exp['CLOUD'], 'DEM': np.full(exp['WTR'].shape, 255, np.uint8)}
    for i, n in enumerate(written):
        assert np.array_equal(stack[i], want[n]), n
    assert not stack[9].any()
    assert info.metadata['OCEAN_MASKING_ENABLED'] == 'TRUE'
    # 'cover' mode goes through the split (dilation) path
    ok = D.generate_dswx_layers(files, mask_adjacent_to_cloud_mode='cover',
                                output_cloud_layer=str(tmp_path / 'cloud_cover.tif'),
                                output_interpreted_band=str(tmp_path / 'wtr_cover.tif'))
    assert ok is True
    exp = o.classify_tile(s['bands'], s['fmask'], mask_adjacent_to_cloud_mode='cover')
    for name, layer in (('cloud_cover.tif', 'CLOUD'), ('wtr_cover.tif', 'WTR')):
        arr, info = geotiff.read_geotiff(str(tmp_path / name))
        assert np.array_equal(arr, exp[layer]), layer
        assert info.metadata['MASK_ADJACENT_TO_CLOUD_MODE'] == 'cover'


def test_offset_and_scale_inputs_flag(tmp_path):
    """flag_offset_and_scale_inputs (CLI --offset-and-scale-inputs, reference :2300-2302): the chain on float32
    reflectances scaled with the bands' own scale_factor / add_offset metadata (0.0001 / 0 in the synthetic files).
    With the default (digital-number) thresholds and with thresholds in reflectance units; the RGB composite is the
    same file either way (:3013)."""
    _, files, _, s = synth_hls.make(str(tmp_path), size=200, tile=6)
    scale = [(0.0001, 0.0)] * 6
    refl = dict(pswt_1_nir=0.15, pswt_1_swir1=0.09, pswt_2_blue=0.1, pswt_2_nir=0.25, pswt_2_swir1=0.3,
                pswt_2_swir2=0.1, lcmask_nir=0.12)
    for tag, thr_kw in (('dn', {}), ('refl', refl)):
        thr = D.HlsThresholds()
        base = o.Thresholds(**thr_kw)
        for k in _capi.THRESHOLD_NAMES:
            setattr(thr, k, getattr(base, k))
        outs = {n: str(tmp_path / f'{tag}_{n}.tif') for n in ('wtr', 'diag', 'conf', 'rgb')}
        ok = D.generate_dswx_layers(files, hls_thresholds=thr, flag_offset_and_scale_inputs=True,
                                    output_interpreted_band=outs['wtr'], output_diagnostic_layer=outs['diag'],
                                    output_confidence_layer=outs['conf'], output_rgb_file=outs['rgb'])
        assert ok is True
        exp = o.classify_tile(s['bands'], s['fmask'], base, offset_and_scale=scale)
        for n, layer in (('wtr', 'WTR'), ('diag', 'DIAG'), ('conf', 'CONF')):
            arr, _ = geotiff.read_geotiff(outs[n])
            assert np.array_equal(arr, exp[layer]), (tag, layer)
        rgb, _ = geotiff.read_geotiff(outs['rgb'])
        want = np.float32(0.0001) * (np.clip(s['bands'][2], 1, None).astype(np.float32) - np.float32(0.0))
        valid = exp['DIAG'] != 65535
        assert rgb.dtype == np.float32 and np.array_equal(rgb[0][valid], want[valid]) and np.isnan(rgb[0][~valid]).all()
    # the flag changes the result with the default thresholds (every `band < threshold` test turns true)
    plain = o.classify_tile(s['bands'], s['fmask'])
    assert not np.array_equal(plain['DIAG'], o.classify_tile(s['bands'], s['fmask'], offset_and_scale=scale)['DIAG'])


def test_browse_png_and_rgb_outputs(tmp_path):
    import struct
    import zlib
    _, files, _, s = synth_hls.make(str(tmp_path), size=256, tile=2)
    ok = D.generate_dswx_layers(
        files, output_browse_image=str(tmp_path / 'b.png'), browse_image_height=64,
        browse_image_width=32, output_rgb_file=str(tmp_path / 'rgb.tif'),
        output_infrared_rgb_file=str(tmp_path / 'irgb.tif'), not_water_in_browse='nodata')
    assert ok is True
    raw = o.classify_tile(s['bands'], s['fmask'], collapse=False)
    exp_browse = o.compute_browse_array(raw['WTR'], True, True, set_not_water_to_nodata=True)
    btif, info = geotiff.read_geotiff(str(tmp_path / 'b.tif'))
    assert np.array_equal(btif, exp_browse) and info.nodata == 255
    assert info.colormap[1].tolist() == [0, 0, 255] and info.colormap[253].tolist() == [175, 175, 175]
    png = (tmp_path / 'b.png').read_bytes()
    assert png[:8] == b'\x89PNG\r\n\x1a\n'
    w, h, depth, ctype = struct.unpack('>IIBB', png[16:26])
    assert (w, h, depth, ctype) == (32, 64, 8, 3)
    pos, idat = 8, b''
    while pos < len(png):
        n, tag = struct.unpack('>I4s', png[pos:pos + 8])
        if tag == b'IDAT':
            idat += png[pos + 8:pos + 8 + n]
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(64, 33)
    assert np.array_equal(rows[:, 1:], geotiff.resample_nearest(exp_browse, 64, 32))
    # composites: scale * (float32(clipped band) - offset), NaN where invalid
    rgb, info = geotiff.read_geotiff(str(tmp_path / 'rgb.tif'))
    invalid = raw['DIAG'] == 65535
    for i, k in enumerate((2, 1, 0)):        # red, green, blue
        exp = 0.0001 * (np.clip(s['bands'][k], 1, None).astype(np.float32) - 0.0)
        exp = exp.astype(np.float32)
        exp[invalid] = np.nan
        assert rgb.dtype == np.float32 and np.array_equal(rgb[i], exp, equal_nan=True)
    irgb, _ = geotiff.read_geotiff(str(tmp_path / 'irgb.tif'))
    exp = (0.0001 * (np.clip(s['bands'][4], 1, None).astype(np.float32) - 0.0)).astype(np.float32)
    exp[invalid] = np.nan
    assert np.array_equal(irgb[0], exp, equal_nan=True)


def test_batch_driver_single_gpu(tmp_path):
    """Node-level driver with one GPU: three tiles through one worker process."""
    from proteus_amd import batch
    rcs = []
    for t in range(3):
        d = tmp_path / f'tile{t}'
        rcfile, _, _, _ = synth_hls.make(str(d), size=128, tile=20 + t, product_id=f'T{t}')
        rcs.append(rcfile)
    ok, results = batch.run_batch(rcs, 1)
    assert ok, results
    assert [r['runconfig'] for r in results] == rcs and all(r['device'] == 0 for r in results)
    for t in range(3):
        arr, _ = geotiff.read_geotiff(str(tmp_path / f'tile{t}' / 'output' / f'T{t}_v1.0_B01_WTR.tif'))
        s = synth_hls.synth_tile(20 + t, 128, 128)
        assert np.array_equal(arr, o.classify_tile(s['bands'], s['fmask'])['WTR'])
    # several workers on the one GPU (overlapping the host-side codec of different tiles)
    ok, results = batch.run_batch(rcs, 1, workers_per_gpu=3)
    assert ok and [r['runconfig'] for r in results] == rcs and all(r['device'] == 0 for r in results)
    # resume: nothing is recomputed when every requested output exists; a deleted layer is
    ok, results = batch.run_batch(rcs, 1, skip_existing=True)
    assert ok and all(r.get('skipped') for r in results)
    os.remove(str(tmp_path / 'tile1' / 'output' / 'T1_v1.0_B03_CONF.tif'))
    ok, results = batch.run_batch(rcs, 1, skip_existing=True)
    assert ok and [bool(r.get('skipped')) for r in results] == [True, False, True]
    assert (tmp_path / 'tile1' / 'output' / 'T1_v1.0_B03_CONF.tif').exists()
    # a broken runconfig is reported, the others still run
    ok, results = batch.run_batch([rcs[0], str(tmp_path / 'missing.yaml')], 1)
    assert not ok and results[0]['ok'] and not results[1]['ok']


def test_product_from_band_files_written_by_another_library(tmp_path):
    """The reader side of the pipeline on FOREIGN layouts: the seven band files rewritten by Pillow / libtiff -- DEFLATE
    STRIPS with the horizontal predictor (the short last strip; the device's untile kernel with block width = raster
    width, not a multiple of 8), one of them uncompressed with a stray PREDICTOR tag (libtiff ignores it there), two of them
    LZW files (the native codec's LZW decoder) -- carrying
    the GDAL metadata and nodata tags of the originals.  The product's layers are what the oracle computes from the
    arrays."""
    from PIL import Image, TiffImagePlugin, features
    if not features.check('libtiff'):
        pytest.skip('Pillow without libtiff')
    from proteus_amd import dswx_hls as D
    size = 333
    rcfile, files, _, s = synth_hls.make(str(tmp_path / 'own'), size=size, tile=61)
    foreign = tmp_path / 'foreign'
    foreign.mkdir()
    new_files = []
    for k, path in enumerate(files):
        arr, info = geotiff.read_geotiff(path)
        ifd = TiffImagePlugin.ImageFileDirectory_v2()
        ifd[42112] = geotiff._metadata_xml(info.metadata, None)
        ifd.tagtype[42112] = 2
        if info.nodata is not None:
            ifd[42113] = str(int(info.nodata))
            ifd.tagtype[42113] = 2
        ifd[317] = 2
        if arr.dtype == np.int16:
            ifd[339] = 2                                        # SampleFormat: two's complement
        dst = str(foreign / os.path.basename(path))
        img = Image.fromarray(arr.view(np.uint16) if arr.dtype == np.int16 else arr)
        img.save(dst, compression=None if k == 2 else ('tiff_lzw' if k in (4, 6) else 'tiff_adobe_deflate'), tiffinfo=ifd)
        d = geotiff.open_geotiff(dst)
        assert not d.tiled and d.bw == size and d.info.dtype == arr.dtype and d.predictor == (1 if k == 2 else 2)
        assert d.comp == (1 if k == 2 else (5 if k in (4, 6) else 8))
        assert k == 2 or (d.down > 1 and size % d.bh)           # DEFLATE: several strips, the last one short
        new_files.append(dst)
    out = {n: str(tmp_path / f'{n}.tif') for n in ('wtr', 'conf', 'diag', 'cloud')}
    assert D.generate_dswx_layers(new_files, output_interpreted_band=out['wtr'], output_confidence_layer=out['conf'],
                                  output_diagnostic_layer=out['diag'], output_cloud_layer=out['cloud'],
                                  scratch_dir=str(tmp_path / 'scratch'))
    exp = o.classify_tile(s['bands'], s['fmask'])
    for n, layer in (('wtr', 'WTR'), ('conf', 'CONF'), ('diag', 'DIAG'), ('cloud', 'CLOUD')):
        arr, _ = geotiff.read_geotiff(out[n])
        assert np.array_equal(arr, exp[layer]), layer


def test_a_damaged_band_file_is_an_error_return_not_a_crash(tmp_path):
    """The reference: gdal.Open fails -> 'ERROR could not open' -> generate_dswx_layers returns False (:4988-4990).  Here a
    band file with corrupt block data (the native codec reports it), one with a damaged directory, and a truncated one
    each make the run return False; the next run on the intact files succeeds (nothing is left in a bad state)."""
    from proteus_amd import dswx_hls as D
    _, files, _, s = synth_hls.make(str(tmp_path / 'ok'), size=200, tile=77)
    good = open(files[3], 'rb').read()
    out = str(tmp_path / 'wtr.tif')
    for kind in ('data', 'directory', 'truncated'):
        buf = bytearray(good)
        if kind == 'data':
            buf[-2000:-1000] = bytes(1000)
        elif kind == 'directory':
            buf[20:24] = (0xfffffff0).to_bytes(4, 'little')
        else:
            buf = buf[:len(buf) // 2]
        open(files[3], 'wb').write(bytes(buf))
        assert D.generate_dswx_layers(files, output_interpreted_band=out, scratch_dir=str(tmp_path / 'scratch')) is False, kind
        assert not os.path.exists(out)
    open(files[3], 'wb').write(good)
    assert D.generate_dswx_layers(files, output_interpreted_band=out, scratch_dir=str(tmp_path / 'scratch')) is True
    arr, _ = geotiff.read_geotiff(out)
    assert np.array_equal(arr, o.classify_tile(s['bands'], s['fmask'])['WTR'])


def test_tiles_in_flight_inside_one_worker(tmp_path):
    """Round 6: ONE worker process keeps several tiles going at once on its threads -- tile k + 1
    inflating and tile k - 1 deflating while tile k is on the device, the engine's lock serialising the GPU part, one HIP
    context.  Eight tiles of different sizes, sensors and ancillary sets, six in flight: every layer of every product is
    what the same tile gives alone (oracle), every file is a valid COG, and the worker's stage report says the stages
    overlapped."""
    from proteus_amd import batch
    rcs, want = [], []
    for t in range(8):
        size = (96, 160, 131, 64)[t % 4]
        d = tmp_path / f'f{t}'
        anc = t % 3 == 0
        rcfile, _, _, _ = synth_hls.make(str(d), sensor=('L30', 'S30')[t % 2], size=size, tile=50 + t, product_id=f'F{t}',
                                         ancillary=anc, ocean=anc)
        rcs.append(rcfile)
        want.append((size, anc))
    reports = []
    ok, results = batch.run_batch(rcs, 1, in_flight=6, reports=reports)
    assert ok, results
    assert len(reports) == 1 and reports[0]['in_flight'] == 6 and reports[0]['tiles'] == 8
    st = reports[0]['stages']['stages']
    assert st['gpu: classify (resident planes)']['spans'] == 8 and st['write: deflate']['spans'] >= 8 * 7
    for t, (size, anc) in enumerate(want):
        out = tmp_path / f'f{t}' / 'output'
        s = synth_hls.synth_tile(50 + t, size, size)
        if anc:
            continue                    # (ancillary products: test_mixed_stream_... checks their layers; here: they ran side by side)
        exp = o.classify_tile(s['bands'], s['fmask'])
        for stem, layer in (('B01_WTR', 'WTR'), ('B02_BWTR', 'BWTR'), ('B03_CONF', 'CONF'), ('B04_DIAG', 'DIAG'),
                            ('B05_WTR-1', 'WTR-1'), ('B06_WTR-2', 'WTR-2'), ('B09_CLOUD', 'CLOUD')):
            path = str(out / f'F{t}_v1.0_{stem}.tif')
            arr, _ = geotiff.read_geotiff(path)
            assert np.array_equal(arr, exp[layer]), (t, layer)
            assert geotiff.validate_cog(path) == []


def test_mixed_stream_with_terrain_shadow_and_landcover(tmp_path):
    """BASELINE.json configs[4] on one GPU: a mixed HLS.L30 / HLS.S30 stream through the node-level
    driver, from runconfigs alone, with terrain shadow + ocean masking + land cover enabled: SHAD computed
    from a DEM and LAND from CGLS + WorldCover (both GPU kernels), the ocean mask from the shoreline input
    (rasters already on the product grid); WTR-2 / CONF / WTR / DIAG / SHAD / LAND / DEM validated against
    the oracle.  512 x 512 tiles: more than one dilation / shadow block in each direction."""
    from proteus_amd import batch
    from proteus_amd.synth import synth_dem, synth_landcover_inputs
    size, rcs = 512, []
    for t in range(4):
        rcfile, _, _, _ = synth_hls.make(str(tmp_path / f'tile{t}'), sensor=('L30', 'S30')[t % 2], size=size,
                                         tile=40 + t, product_id=f'M{t}', ancillary=True, ocean=True)
        rcs.append(rcfile)
    ok, results = batch.run_batch(rcs, 1)
    assert ok, results
    forest = [20, 50, 111, 113, 115, 116, 121, 123, 125, 126]     # defaults/dswx_hls.yaml
    for t in range(4):
        s = synth_hls.synth_tile(40 + t, size, size, with_masks=True)
        dem = synth_dem(40 + t, size + 100, size + 100)
        # sun angles of the synthetic product: azimuth 143.2, zenith 34.5 (tools/make_synthetic_hls.py)
        # the host mirror defaults to the promotion of the numpy the reference pins (1.23.5: value-based casting)
        shad = o.compute_opera_shadow_layer(dem, 143.2, 90 - 34.5, -5, 40, legacy_promotion=True)[50:-50, 50:-50]
        wc, cg = synth_landcover_inputs(40 + t, size, size)
        land = o.landcover_mask_from_warped(wc, cg, forest, year=2021)
        exp = o.classify_tile(s['bands'], s['fmask'], landcover=land, shadow=shad, ocean_mask=s['ocean'], collapse=True)
        out = tmp_path / f'tile{t}' / 'output'
        read = lambda stem: geotiff.read_geotiff(str(out / f'M{t}_v1.0_{stem}.tif'))   # noqa: E731
        assert np.array_equal(read('B08_SHAD')[0], shad.astype(np.uint8))
        assert np.array_equal(read('B07_LAND')[0], land)
        demr, info = read('B10_DEM')
        assert demr.dtype == np.float32 and np.array_equal(demr, dem[50:-50, 50:-50]) and np.isnan(info.nodata)
        for layer, stem in (('WTR-2', 'B06_WTR-2'), ('CONF', 'B03_CONF'), ('WTR', 'B01_WTR'), ('DIAG', 'B04_DIAG')):
            arr, info = read(stem)
            assert np.array_equal(arr, exp[layer]), (t, layer)
        md = info.metadata
        assert md['DEM_SOURCE'] == 'Synthetic DEM' and md['WORLDCOVER_SOURCE'] == 'Synthetic ESA WorldCover 10m 2021'
        assert md['SPACECRAFT_NAME'] == ('Landsat-8', 'Sentinel-2A')[t % 2]
        assert md['OCEAN_MASKING_ENABLED'] == 'TRUE' and md['SHORELINE_SOURCE'] == 'Synthetic shoreline raster'
        # the shadow, land-cover and ocean rules did change pixels, or the test is vacuous
        plain = o.classify_tile(s['bands'], s['fmask'], collapse=True)
        assert not np.array_equal(plain['WTR-2'], exp['WTR-2'])
        assert (exp['WTR'] == 254).any() and not (plain['WTR'] == 254).any()


def test_ancillary_files_off_grid_are_refused(tmp_path):
    """Rasters that would need gdal.Warp are refused loudly, not resampled silently."""
    rcfile, files, _, _ = synth_hls.make(str(tmp_path), size=64, ancillary=True, dem_margin=50)
    bad = str(tmp_path / 'dem_shifted.tif')
    dem, info = geotiff.read_geotiff(str(tmp_path / 'ancillary' / 'dem.tif'))
    gt = list(info.geotransform)
    gt[0] += 7.0                                      # off the grid by a fraction of a pixel
    geotiff.write_geotiff(bad, dem, geo_tags=geotiff.geo_tags_from_geotransform(tuple(gt), 32615))
    with pytest.raises(NotImplementedError, match='not on the HLS grid'):
        D.generate_dswx_layers(files, dem_file=bad, scratch_dir=str(tmp_path))
    with pytest.raises(NotImplementedError, match='otsu'):
        D.generate_dswx_layers(files, dem_file=str(tmp_path / 'ancillary' / 'dem.tif'),
                               shadow_masking_algorithm='otsu', scratch_dir=str(tmp_path))
    with pytest.raises(NotImplementedError, match='shoreline'):
        D.generate_dswx_layers(files, shoreline_shapefile='coast.shp', apply_ocean_masking=True,
                               scratch_dir=str(tmp_path))


def test_product_option_fuzz_and_inflight_soak():
    """Short runs of the two product-path soaks in child processes (their long runs: profiles/r06_product_fuzz.json,
    r06_inflight_soak.json): random option sets through generate_dswx_layers with four runs side by side in one process, and
    the node-level driver with six tiles in flight -- every output file checked against the oracles inside the helpers."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for argv in (['product_fuzz.py', '--cases', '48', '--threads', '4', '--seed', '9'],
                 ['inflight_soak.py', '--tiles', '24', '--in-flight', '6']):
        res = subprocess.run([sys.executable, os.path.join(root, 'tests', 'helpers', argv[0])] + argv[1:], capture_output=True,
                             text=True, timeout=600)
        assert res.returncode == 0, (argv, res.stdout[-800:], res.stderr[-800:])
        assert json.loads(res.stdout.strip().splitlines()[-1])['ok'] is True


def test_bench_product_run_leg(monkeypatch):
    """bench.py's `product_run` record (the plain N = 1 command, after the timed regions): one product of the synthetic
    recipe and one of a coherent scene in child processes -- at a small tile size here."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    monkeypatch.setattr(bench, 'TILE', 300)
    rec = bench.product_run_leg(timeout_s=200)
    for key in ('synthetic_recipe_tile', 'coherent_scene_tile'):
        assert 'error' not in rec[key], rec[key]
        assert rec[key]['seconds_per_product'] > 0 and rec[key]['stages_wall_s']['gpu: classify (resident planes)'] >= 0
        assert rec[key]['output_MB'] > 0
    # a child that cannot run is a record, not an exception
    monkeypatch.setattr(bench, 'TILE', 0)
    assert 'error' in bench.product_run_leg(timeout_s=60)['synthetic_recipe_tile']


def test_nodata_wedge_and_fmask_fill_over_valid_reflectances(tmp_path):
    """A granule at a swath edge: the top 45 % of the tile is nodata (Fmask 255, bands -9999), and a rectangle further down
    carries Fmask fill over VALID reflectances (the reference calls such pixels invalid all the same: its cumulative fill
    test includes Fmask, dswx_hls.py:2195-2209); band files in 128 x 128 tiles.  Every layer, the composite and the coverage
    metadata are what the oracle computes from the arrays that were written; without a usable Fmask fill value (no nodata
    tag, no _FillValue) the Fmask plane invalidates nothing, as in the reference."""
    from proteus_amd import dswx_hls as D
    size = 700
    rcfile, files, _, s = synth_hls.make(str(tmp_path / 'in'), size=size, tile=88, fill_rows=0.45, file_tile=128)
    fm = np.array(s['fmask'])
    fm[400:660, 130:520] = 255
    geo, meta = geotiff.read_geotiff(files[0])[1].geo_tags, geotiff.read_geotiff(files[-1])[1].metadata
    geotiff.write_geotiff(files[-1], fm, geo_tags=geo, metadata=meta, nodata=255, tile=128)
    outs = {n: str(tmp_path / f'{n}.tif') for n in ('wtr', 'conf', 'diag', 'cloud', 'rgb', 'wtr1', 'wtr2', 'bwtr')}
    assert D.generate_dswx_layers(files, output_interpreted_band=outs['wtr'], output_confidence_layer=outs['conf'],
                                  output_diagnostic_layer=outs['diag'], output_cloud_layer=outs['cloud'],
                                  output_rgb_file=outs['rgb'], output_non_masked_dswx=outs['wtr1'],
                                  output_shadow_masked_dswx=outs['wtr2'], output_binary_water=outs['bwtr'],
                                  scratch_dir=str(tmp_path / 'scratch'))
    exp = o.classify_tile(s['bands'], fm)
    for n, layer in (('wtr', 'WTR'), ('conf', 'CONF'), ('diag', 'DIAG'), ('cloud', 'CLOUD'), ('wtr1', 'WTR-1'), ('wtr2', 'WTR-2'),
                     ('bwtr', 'BWTR')):
        arr, info = geotiff.read_geotiff(outs[n])
        assert np.array_equal(arr, exp[layer]), layer
    assert info.metadata['SPATIAL_COVERAGE'] == str(exp['counters']['SPATIAL_COVERAGE'])
    rgb, _ = geotiff.read_geotiff(outs['rgb'])
    valid = exp['DIAG'] != 65535
    want = np.float32(0.0001) * (np.clip(s['bands'][2], 1, None).astype(np.float32) - np.float32(0.0))
    assert np.isnan(rgb[:, ~valid]).all() and np.array_equal(rgb[0][valid], want[valid]) and not valid[:300].any()
    assert not valid[400:660, 130:520].any() and (s['bands'][2][400:660, 130:520] != -9999).any()
    geotiff.write_geotiff(files[-1], fm, geo_tags=geo, metadata={k: v for k, v in meta.items() if k != '_FillValue'}, tile=128)
    assert D.generate_dswx_layers(files, output_interpreted_band=outs['wtr'], scratch_dir=str(tmp_path / 'scratch'))
    arr, _ = geotiff.read_geotiff(outs['wtr'])
    assert np.array_equal(arr, o.classify_tile(s['bands'], fm, fmask_fill=-9999.)['WTR'])
