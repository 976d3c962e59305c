"""BASELINE.json configs[0]/[1] plumbing on the GPU: synthetic HLS GeoTIFFs + runconfig
-> bin/dswx_hls.py -> GeoTIFF layers, checked against the oracle; plus the reference's own
unit test (tests/test_dswx_hls_units.py) restated against this package."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import dswx_oracle as o
from proteus_amd import dswx_hls as D
from proteus_amd import geotiff

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
    # bands in band_description_dict order, Byte layers only; WTR-1 is the post-aerosol one
    names = ['WTR', 'BWTR', 'CONF', 'WTR-1', 'WTR-2', 'LAND', 'SHAD', 'CLOUD']
    assert info.descriptions == [D.band_description_dict[n] for n in names]
    want = {'WTR': exp['WTR'], 'BWTR': exp['BWTR'], 'CONF': exp['CONF'],
            'WTR-1': exp['WTR-1-AEROSOL'], 'WTR-2': exp['WTR-2'], 'LAND': s['land'],
            'SHAD': s['shad'], 'CLOUD': exp['CLOUD']}
    for i, n in enumerate(names):
        assert np.array_equal(stack[i], want[n]), n
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
    # a broken runconfig is reported, the others still run
    ok, results = batch.run_batch([rcs[0], str(tmp_path / 'missing.yaml')], 1)
    assert not ok and results[0]['ok'] and not results[1]['ok']
