"""Host side of the MI355X DSWx-HLS drop-in: the same public names as PROTEUS
`src/proteus/dswx_hls.py`, with the per-pixel chain running on the GPU.

Kept from the reference interface (file:line = src/proteus/dswx_hls.py):
  get_dswx_hls_cli_parser :411      parse_runconfig_file :3601    create_logger :4126
  generate_dswx_layers :4610        compare_dswx_hls_products :710
  HlsThresholds :274                RunConfigConstants :321
  interpreted_dswx_band_dict :97    generate_interpreted_layer :1687
  band_description_dict :217        layer_names_to_args_dict :229  collapse_wtr_classes_dict :201

What changes: the run of numpy calls inside generate_dswx_layers (:5089, :5110-5112,
:5225-5231, :5245-5249, :5261, :5268, :5282, :5286, :5358, :5368) is ONE call into the HIP
library (proteus_amd._capi.Context.classify_host -> dswx_classify_host).  There is no
numpy implementation of that chain in this package: without the built library or
without a GPU, generate_dswx_layers raises.

What stays on the host and is NOT re-implemented here (out of scope):
GDAL reprojection of DEM / land cover / shoreline, hillshade ('otsu' shadow masking) and HDF4
input.  GDAL is not installed in this image, so raster I/O goes through proteus_amd.geotiff.
Ancillary FILES are accepted when they are already on the product grid -- `dem_file` on the HLS
grid with a margin (-> GPU terrain-shadow kernel, SHAD and DEM layers), `landcover_file` on the
HLS grid and `worldcover_file` on the grid three times finer (-> GPU LAND aggregation) -- i.e.
what the reference's own `_warp` calls would hand to its per-pixel code; anything that still
needs warping, and `shoreline_shapefile`, raises NotImplementedError.  Pre-gridded planes can
also be handed over directly with the `landcover_mask=`, `shadow_layer=` and `ocean_mask=`
keyword extensions (arrays or GeoTIFF paths).
"""
import argparse
import glob
import logging
import os
import sys
import threading
from collections import OrderedDict
from datetime import datetime

import numpy as np

from . import _capi, geotiff, pipeline, stages, runconfig as _rc
from .version import VERSION as SOFTWARE_VERSION

FLAG_COLLAPSE_WTR_CLASSES = True          # :26
FLAG_CLIP_NEGATIVE_REFLECTANCE = True     # :31
SCALE_FACTOR = 0.0001                     # :44
AEROSOL_REMAPPING_MAX_NIR = 0.1 / SCALE_FACTOR
COMPARE_DSWX_HLS_PRODUCTS_ERROR_TOLERANCE = 1e-6
UINT8_FILL_VALUE = 255
OCEAN_MASKED_RGBA = (0, 0, 127, 0)
FILL_VALUE_RGBA = (0, 0, 0, 0)
DIAGNOSTIC_LAYER_NO_DATA_DECIMAL = 0b100000
DIAGNOSTIC_LAYER_NO_DATA_BINARY_REPR = 65535
WTR_SNOW_MASKED, WTR_CLOUD_MASKED, WTR_OCEAN_MASKED = 252, 253, 254

logger = logging.getLogger('dswx_hls')

# HLS v2 per-band file suffixes (:78-92); v1 (HDF4, :62-76) needs GDAL
l30_v2_band_dict = {'blue': 'B02', 'green': 'B03', 'red': 'B04', 'nir': 'B05',
                    'swir1': 'B06', 'swir2': 'B07', 'fmask': 'Fmask'}
s30_v2_band_dict = {'blue': 'B02', 'green': 'B03', 'red': 'B04', 'nir': 'B8A',
                    'swir1': 'B11', 'swir2': 'B12', 'fmask': 'Fmask'}

# DIAG value -> WTR-1 class (:97-143), grouped by class
interpreted_dswx_band_dict = {}
for _cls, _keys in ((0, (0b00000, 0b00001, 0b00010, 0b00100, 0b01000)),
                    (1, (0b01111, 0b10111, 0b11011, 0b11101, 0b11110, 0b11111)),
                    (2, (0b00111, 0b01011, 0b01101, 0b01110, 0b10011, 0b10101, 0b10110,
                         0b11001, 0b11010, 0b11100)),
                    (3, (0b11000,)),
                    (4, (0b00011, 0b00101, 0b00110, 0b01001, 0b01010, 0b01100, 0b10000,
                         0b10001, 0b10010, 0b10100))):
    for _k in _keys:
        interpreted_dswx_band_dict[_k] = _cls
interpreted_dswx_band_dict[DIAGNOSTIC_LAYER_NO_DATA_DECIMAL] = UINT8_FILL_VALUE

collapse_wtr_classes_dict = {0: 0, 1: 1, 2: 1, 3: 2, 4: 2, WTR_OCEAN_MASKED: WTR_OCEAN_MASKED,
                             WTR_SNOW_MASKED: WTR_SNOW_MASKED,
                             WTR_CLOUD_MASKED: WTR_CLOUD_MASKED,
                             UINT8_FILL_VALUE: UINT8_FILL_VALUE}
collapsable_layers_list = ['WTR', 'WTR-1', 'WTR-2']

band_description_dict = OrderedDict([
    ('WTR', 'Water classification (WTR)'),
    ('BWTR', 'Binary Water (BWTR)'),
    ('CONF', 'Confidence classification (CONF)'),
    ('DIAG', 'Diagnostic layer (DIAG)'),
    ('WTR-1', 'Interpretation of diagnostic layer into water classes (WTR-1)'),
    ('WTR-2', 'Interpreted layer refined using land cover and terrain shadow testing (WTR-2)'),
    ('LAND', 'Land cover classification (LAND)'),
    ('SHAD', 'Terrain shadow layer (SHAD)'),
    ('CLOUD', 'Input HLS Fmask cloud/cloud-shadow classification (CLOUD)'),
    ('DEM', 'Digital elevation model (DEM)')])

layer_names_to_args_dict = OrderedDict([
    ('WTR', 'output_interpreted_band'), ('BWTR', 'output_binary_water'),
    ('CONF', 'output_confidence_layer'), ('DIAG', 'output_diagnostic_layer'),
    ('WTR-1', 'output_non_masked_dswx'), ('WTR-2', 'output_shadow_masked_dswx'),
    ('LAND', 'output_landcover'), ('SHAD', 'output_shadow_layer'),
    ('CLOUD', 'output_cloud_layer'), ('DEM', 'output_dem_layer'),
    ('RGB', 'output_rgb_file'), ('INFRARED_RGB', 'output_infrared_rgb_file')])

METADATA_FIELDS_TO_COPY_FROM_HLS_LIST = ['MEAN_SUN_AZIMUTH_ANGLE', 'MEAN_SUN_ZENITH_ANGLE',
                                         'MEAN_VIEW_AZIMUTH_ANGLE', 'MEAN_VIEW_ZENITH_ANGLE',
                                         'NBAR_SOLAR_ZENITH', 'ACCODE']

_THRESHOLD_NAMES = _capi.THRESHOLD_NAMES
_AEROSOL_KEYS = (
    'aerosol_not_water_to_high_conf_water_fmask_values',
    'aerosol_water_moderate_conf_to_high_conf_water_fmask_values',
    'aerosol_partial_surface_water_conservative_to_high_conf_water_fmask_values',
    'aerosol_partial_surface_aggressive_to_high_conf_water_fmask_values')


class HlsThresholds:
    """The twelve reflectance thresholds (:274-318)."""

    def __init__(self):
        for name in _THRESHOLD_NAMES:
            setattr(self, name, None)


class RunConfigConstants:
    """Processing / browse constants that come from the runconfig (:321-408)."""

    _FIELDS = ('check_ancillary_inputs_coverage', 'apply_ocean_masking',
               'apply_aerosol_class_remapping') + _AEROSOL_KEYS + (
        'shadow_masking_algorithm', 'min_slope_angle', 'max_sun_local_inc_angle',
        'mask_adjacent_to_cloud_mode', 'forest_mask_landcover_classes',
        'ocean_masking_shoreline_distance_km', 'browse_image_height', 'browse_image_width',
        'exclude_psw_aggressive_in_browse', 'not_water_in_browse', 'cloud_in_browse',
        'snow_in_browse')

    def __init__(self):
        self.hls_thresholds = HlsThresholds()
        for name in self._FIELDS:
            setattr(self, name, None)


# -----------------------------------------------------------------------------------
# command line (:411-702): same flags, same dest names
# -----------------------------------------------------------------------------------
def get_dswx_hls_cli_parser():
    p = argparse.ArgumentParser(
        description='Generate a DSWx-HLS product from an HLS product',
        formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('input_list', type=str, nargs='+',
                   help='Input YAML run configuration file or HLS product file(s)')

    def files(*flags, dest, help):
        p.add_argument(*flags, dest=dest, type=str, help=help)

    files('--dem', dest='dem_file', help='Input digital elevation model (DEM)')
    files('--dem-description', dest='dem_file_description', help='DEM description')
    files('-c', '--landcover', dest='landcover_file',
          help='Input Copernicus Land Cover Discrete-Classification-map 100m')
    files('--landcover-description', dest='landcover_file_description',
          help='Land cover description')
    files('-w', '--worldcover', dest='worldcover_file', help='Input ESA WorldCover 10m')
    files('--worldcover-description', dest='worldcover_file_description',
          help='WorldCover description')
    files('-s', '--shoreline', dest='shoreline_shapefile', help='NOAA GSHHS shapefile')
    files('--shoreline-shape-description', dest='shoreline_shapefile_description',
          help='NOAA GSHHS shapefile description')
    # outputs.  The reference concatenates a few alias pairs by accident (missing
    # commas, e.g. '--bwtr--output-binary-water' at :496-497); both the intended and
    # the accidental spellings are accepted.
    files('-o', '--output-file', dest='output_file', help='Output DSWx-HLS product (GeoTIFF)')
    files('--wtr', '--interpreted-band', dest='output_interpreted_band',
          help='Output interpreted DSWx layer (GeoTIFF)')
    files('--output-rgb', '--output-rgb-file', dest='output_rgb_file',
          help='Output RGB reflectance file (GeoTIFF)')
    files('--output-infrared-rgb', '--output-infrared-rgb-file',
          dest='output_infrared_rgb_file', help='Output SWIR-1/NIR/Red composition (GeoTIFF)')
    files('--bwtr', '--output-binary-water', '--bwtr--output-binary-water',
          dest='output_binary_water', help='Output binary water mask (GeoTIFF)')
    files('--conf', '--output-confidence-layer', '--conf--output-confidence-layer',
          dest='output_confidence_layer', help='Output confidence layer (GeoTIFF)')
    files('--diag', '--output-diagnostic-layer', dest='output_diagnostic_layer',
          help='Output diagnostic test layer file (GeoTIFF)')
    files('--wtr-1', '--output-non-masked-dswx', dest='output_non_masked_dswx',
          help='Output non-masked DSWx layer file (GeoTIFF)')
    files('--wtr-2', '--output-shadow-masked-dswx', dest='output_shadow_masked_dswx',
          help='Output interpreted layer refined with land cover and terrain shadow (GeoTIFF)')
    files('--land', '--output-land', dest='output_landcover',
          help='Output landcover classification file (GeoTIFF)')
    files('--shad', '--output-shadow-layer', dest='output_shadow_layer',
          help='Output terrain shadow layer file (GeoTIFF)')
    files('--cloud', '--output-cloud-mask', '--cloud--output-cloud-mask',
          dest='output_cloud_layer', help='Output cloud/cloud-shadow classification (GeoTIFF)')
    files('--out-dem', '--output-digital-elevation-model', '--output-elevation-layer',
          '--out-dem--output-digital-elevation-model', dest='output_dem_layer',
          help='Output elevation layer file (GeoTIFF)')
    files('--browse', '--output-browse-image', '--browse--output-browse-image',
          dest='output_browse_image', help='Output browse image file (png)')
    p.add_argument('--bheight', '--browse-image-height', '--bheight--browse-image-height',
                   dest='browse_image_height', type=int, help='Browse PNG height in pixels')
    p.add_argument('--bwidth', '--browse-image-width', '--bwidth--browse-image-width',
                   dest='browse_image_width', type=int, help='Browse PNG width in pixels')
    p.add_argument('--exclude-psw-aggressive-in-browse', dest='exclude_psw_aggressive_in_browse',
                   action='store_true', default=None,
                   help='Exclude the Partial Surface Water Aggressive class in the browse image')
    p.add_argument('--not-water-in-browse', dest='not_water_in_browse', type=str,
                   choices=['white', 'nodata'], default=None, help='Not Water in the browse image')
    p.add_argument('--cloud-in-browse', dest='cloud_in_browse', type=str,
                   choices=['gray', 'nodata'], default=None, help='Cloud in the browse image')
    p.add_argument('--snow-in-browse', dest='snow_in_browse', type=str,
                   choices=['cyan', 'gray', 'nodata'], default=None,
                   help='Snow in the browse image')
    p.add_argument('--offset-and-scale-inputs', dest='flag_offset_and_scale_inputs',
                   action='store_true', default=False,
                   help='Offset and scale HLS inputs before processing')
    files('--scratch-dir', '--temp-dir', '--temporary-dir', dest='scratch_dir',
          help='Scratch (temporary) directory')
    files('--pid', '--product-id', dest='product_id', help='Product ID for the metadata')
    files('--product-version', dest='product_version', help='Product version for the metadata')
    for flag, dest, text in (
            ('--check-ancillary-inputs-coverage', 'check_ancillary_inputs_coverage',
             'Check if ancillary inputs cover entirely the output product'),
            ('--apply-ocean-masking', 'apply_ocean_masking', 'Apply ocean masking'),
            ('--apply-aerosol-masking', 'apply_aerosol_class_remapping', 'Apply aerosol masking')):
        p.add_argument(flag, dest=dest, action='store_true', default=None, help=text)
    p.add_argument('--shadow-masking-algorithm', dest='shadow_masking_algorithm', type=str,
                   choices=['otsu', 'sun_local_inc_angle'], help='Shadow masking algorithm')
    p.add_argument('--min-slope-angle', dest='min_slope_angle', type=float, help='')
    p.add_argument('--max-sun-local-inc-angle', dest='max_sun_local_inc_angle', type=float,
                   help='Maximum local-incidence angle')
    p.add_argument('--mask-adjacent-to-cloud-mode', dest='mask_adjacent_to_cloud_mode', type=str,
                   choices=['mask', 'ignore', 'cover'],
                   help='How areas adjacent to cloud/cloud-shadow are handled')
    p.add_argument('--copernicus-forest-classes', dest='forest_mask_landcover_classes',
                   type=list, help='Copernicus CGLS Land Cover 100m forest classes')
    p.add_argument('--ocean-masking-distance-km', dest='ocean_masking_shoreline_distance_km',
                   type=float, help='Ocean masking distance from shoreline in km')
    p.add_argument('--debug', dest='flag_debug', action='store_true', default=False,
                   help='Activate debug mode')
    # extensions of this drop-in (not in the reference): ancillary layers that are
    # already on the HLS grid, and the GPU to run on
    files('--landcover-mask', dest='landcover_mask', help='LAND layer on the HLS grid (GeoTIFF)')
    files('--shadow-layer', dest='shadow_layer', help='SHAD layer on the HLS grid (GeoTIFF)')
    files('--ocean-mask', dest='ocean_mask',
          help='Ocean mask on the HLS grid, 0 = ocean (GeoTIFF); needs --apply-ocean-masking')
    p.add_argument('--device', dest='device', type=int, default=None, help='GPU index')
    files('--log', '--log-file', dest='log_file', help='Log file')
    p.add_argument('--full-log-format', dest='full_log_formatting', action='store_true',
                   default=False, help='Enable full formatting of log messages')
    return p


# -----------------------------------------------------------------------------------
# logging (:4083-4175)
# -----------------------------------------------------------------------------------
class Logger:
    """File-like object feeding complete lines to a logging.Logger (:4083-4123)."""

    def __init__(self, target, level, prefix=''):
        self.logger, self.level, self.prefix, self.buffer = target, level, prefix, ''

    def write(self, message):
        self.buffer += message
        *lines, self.buffer = self.buffer.split('\n')
        for line in lines:
            if line:
                self.logger.log(self.level, self.prefix + line)

    def flush(self):
        if self.buffer:
            self.logger.log(self.level, self.buffer)
        self.buffer = ''


def create_logger(log_file, full_log_formatting=None):
    """Console (+ file) handlers on the 'dswx_hls' logger and stdout/stderr redirected
    into it, as the reference does (:4126-4175)."""
    logger.setLevel(logging.DEBUG)
    if full_log_formatting:
        fmt = logging.Formatter(
            '%(asctime)s.%(msecs)03d, %(levelname)s, DSWx-HLS, %(module)s, 999999, '
            '%(pathname)s:%(lineno)d, "%(message)s"', '%Y-%m-%d %H:%M:%S')
    else:
        fmt = logging.Formatter('%(message)s')
    console = logging.StreamHandler(sys.__stdout__)
    console.setLevel(logging.DEBUG)
    console.setFormatter(fmt)
    logger.addHandler(console)
    if log_file:
        fh = logging.FileHandler(log_file)
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    sys.stdout = Logger(logger, logging.INFO)
    sys.stderr = Logger(logger, logging.ERROR, prefix='[StdErr] ')
    return logger


# -----------------------------------------------------------------------------------
# runconfig (:3601-3814)
# -----------------------------------------------------------------------------------
def parse_runconfig_file(user_runconfig_file=None, args=None):
    """Load the default runconfig, validate + deep-merge the user's over it, fill a
    RunConfigConstants and (if given) the argparse namespace `args`.  Precedence:
    command line > user runconfig > default runconfig."""
    logger.info(f'Default runconfig file: {_rc.DEFAULT_RUNCONFIG}')
    config = _rc.load_yaml(_rc.DEFAULT_RUNCONFIG)
    if user_runconfig_file is not None:
        if not os.path.isfile(user_runconfig_file):
            msg = f'ERROR invalid file {user_runconfig_file}'
            logger.info(msg)
            raise Exception(msg)
        logger.info(f'Input runconfig file: {user_runconfig_file}')
        user = _rc.load_yaml(user_runconfig_file)
        logger.info(f'Validating runconfig file: {user_runconfig_file}')
        _rc.validate_runconfig(user, user_runconfig_file)
        config = _rc.deep_update(config, user)
    groups = config['runconfig']['groups']
    processing, browse = groups['processing'], groups['browse_image_group']
    consts = RunConfigConstants()
    for group in (processing, browse):
        for key, val in group.items():
            if key in RunConfigConstants._FIELDS:
                setattr(consts, key, val)
    thresholds = groups.get('hls_thresholds')
    if thresholds is not None:
        logger.info('HLS thresholds:')
        for key, val in thresholds.items():
            logger.info(f'     {key}: {val}')
            setattr(consts.hls_thresholds, key, val)
    if args is None:
        return consts

    for key in RunConfigConstants._FIELDS:
        if getattr(args, key, None) is None:
            setattr(args, key, getattr(consts, key))

    input_file_path = groups['input_file_group']['input_file_path']
    anc, paths = groups['dynamic_ancillary_file_group'], groups['product_path_group']
    product_id = paths['product_id'] if paths['product_id'] is not None else 'dswx_hls'
    pv = paths['product_version']
    product_version = SOFTWARE_VERSION if pv is None else f'{pv:.1f}'
    if (input_file_path is not None and len(input_file_path) == 1 and
            os.path.isdir(input_file_path[0])):
        logger.info(f'input HLS files directory: {input_file_path[0]}')
        args.input_list = glob.glob(os.path.join(input_file_path[0], '*.tif'))
    elif input_file_path is not None:
        args.input_list = input_file_path

    from_runconfig = {k: anc[k] for k in (
        'dem_file', 'dem_file_description', 'landcover_file', 'landcover_file_description',
        'worldcover_file', 'worldcover_file_description', 'shoreline_shapefile',
        'shoreline_shapefile_description')}
    from_runconfig.update(scratch_dir=paths['scratch_path'], product_id=product_id,
                          product_version=product_version)
    for name, rc_val in from_runconfig.items():
        cli_val = getattr(args, name, None)
        if cli_val is not None and rc_val is not None:
            logger.warning(f'command line {name} "{cli_val}" has precedence over runconfig'
                           f' {name} "{rc_val}".')
        elif cli_val is None:
            setattr(args, name, rc_val)
    if user_runconfig_file is None:
        return consts

    out_dir = paths['output_dir']
    for number, (layer, arg_name) in enumerate(layer_names_to_args_dict.items(), start=1):
        save = processing['save_' + layer.lower().replace('-', '_')]
        cli_val = getattr(args, arg_name, None)
        default_name = os.path.join(
            out_dir, f'{product_id}_v{product_version}_B{number:02}_{layer}.tif')
        if cli_val is not None and save:
            logger.warning(f'command line {arg_name} "{cli_val}" has precedence over runconfig'
                           f' {arg_name} "{default_name}".')
        elif cli_val is None and save:
            setattr(args, arg_name, default_name)
    if browse['save_browse']:
        default_name = os.path.join(out_dir, f'{product_id}_v{product_version}_BROWSE.png')
        cli_val = getattr(args, 'output_browse_image', None)
        if cli_val is not None:
            logger.warning(f'command line output_browse_image "{cli_val}" has precedence over'
                           f' default output_browse_image "{default_name}".')
        else:
            args.output_browse_image = default_name
    return consts


# -----------------------------------------------------------------------------------
# GPU context (one per process and device)
# -----------------------------------------------------------------------------------
_contexts = {}
_contexts_lock = threading.Lock()


def get_context(device=None):
    """The process-wide HIP context for `device` (default: $DSWX_DEVICE or 0).
    Raises if the library is not built or no MI355X is visible: no CPU fallback."""
    if device is None:
        device = int(os.environ.get('DSWX_DEVICE', '0'))
    ctx = _contexts.get(device)
    if ctx is None:
        with _contexts_lock:            # (several tiles may be in flight on threads of one process: proteus_amd.batch)
            ctx = _contexts.get(device)
            if ctx is None:
                ctx = _contexts[device] = _capi.Context(device)
    return ctx


def generate_interpreted_layer(diagnostic_layer):
    """DIAG (decimal) -> WTR-1 classes on the GPU (:1687-1707); anything that is not a
    key of interpreted_dswx_band_dict maps to 255."""
    logger.info('interpreting diagnostic tests (DIAG -> WTR-1)')
    return get_context().interpret_layer(np.asarray(diagnostic_layer))


DEM_MARGIN_IN_PIXELS = 50                 # :58


def _compute_opera_shadow_layer(dem, sun_azimuth_angle, sun_elevation_angle,
                                min_slope_angle, max_sun_local_inc_angle,
                                pixel_spacing_x=30, pixel_spacing_y=30, margin=0, numpy_promotion=None):
    """Terrain shadow mask from sun local-incidence and back-slope angles on the GPU
    (:4215-4283); True = not shadow, False = shadow, same shape as `dem` (or cropped by
    `margin` on all sides, fusing _crop_2d_array_all_sides :4320).  The five float64 sun
    scalars are formed here with numpy exactly as the reference forms them (:4246-4253,
    :4276-4277) and handed to the kernel."""
    dem = np.asarray(dem)
    if dem.ndim != 2 or min(dem.shape) < 2:
        raise ValueError('Shape of array too small to calculate a numerical gradient, '
                         'at least 2 elements are required.')
    sun, sin_az, cos_az, legacy = _shadow_geometry(sun_azimuth_angle, sun_elevation_angle, numpy_promotion)
    if margin < 2 and dem.size >= 1 << 20:
        # the only geometry left on the general one-pixel kernel (border pixels take one-sided differences there):
        # ~0.2 of the HBM rate instead of ~0.65 -- the reference's own call has a margin of 50 (:58, :5161-5167)
        logger.warning(f'WARNING terrain shadow layer with a margin of {margin} pixel(s): the general kernel '
                       '(dswx_shadow_v2) computes the whole raster, about three times slower than the filter kernel')
    return get_context().shadow_layer(
        dem, sun, sin_az, cos_az, min_slope_angle, max_sun_local_inc_angle, pixel_spacing_x, pixel_spacing_y,
        margin=margin, float32=legacy)


def _shadow_geometry(sun_azimuth_angle, sun_elevation_angle, numpy_promotion=None):
    """The five float64 sun scalars, formed with numpy exactly as the reference forms them (:4246-4253, :4276-4277), and
    which numpy the caller wants to agree with: 'legacy' (DEFAULT: numpy < 2 value-based casting, what the reference
    computes in its supported environment -- it pins numpy==1.23.5, setup.py:78 -- where the float64 sun scalars do not
    upcast the float32 DEM arrays) or 'nep50' (numpy >= 2: what the same source computes under a current numpy; the
    committed shadow_s_*.npz goldens were generated by importing the reference under numpy 2.2 and pin this mode).
    env DSWX_NUMPY_PROMOTION overrides.  -> (sun vector, sin az, cos az, legacy?)"""
    sun_azimuth = np.radians(sun_azimuth_angle)
    sun_zenith = np.radians(90 - sun_elevation_angle)
    target_to_sun_unit_vector = [np.sin(sun_azimuth) * np.sin(sun_zenith),
                                 np.cos(sun_azimuth) * np.sin(sun_zenith),
                                 np.cos(sun_zenith)]
    mode = (numpy_promotion or os.environ.get('DSWX_NUMPY_PROMOTION', 'legacy')).lower()
    if mode not in ('nep50', 'legacy'):
        raise ValueError(f"numpy_promotion must be 'nep50' or 'legacy', not {mode!r}")
    return target_to_sun_unit_vector, np.sin(sun_azimuth), np.cos(sun_azimuth), mode == 'legacy'


def _crop_2d_array_all_sides(input_2d_array, margin):
    return input_2d_array[margin:-margin, margin:-margin]


# -----------------------------------------------------------------------------------
# HLS loading (:2136-2425), GDAL-free
# -----------------------------------------------------------------------------------
def _harvest_hls_metadata(meta, md):
    """First reflectance band: copy angles / coverage / ids, derive SPACECRAFT_NAME and
    SENSOR (:2228-2291).  Returns False (after logging ERROR) for unsupported platforms."""
    for k, v in meta.items():
        ku = k.upper()
        if ku in METADATA_FIELDS_TO_COPY_FROM_HLS_LIST:
            md[ku] = v
        elif ku in ('SPATIAL_COVERAGE', 'CLOUD_COVERAGE'):
            md['INPUT_HLS_PRODUCT_' + ku] = v
        elif ku in ('LANDSAT_PRODUCT_ID', 'PRODUCT_URI'):
            md['SENSOR_PRODUCT_ID'] = v
        elif ku == 'SENSING_TIME':
            md['SENSING_TIME'] = v
    sensor = None
    if 'SPACECRAFT_NAME' in meta:
        spacecraft = meta['SPACECRAFT_NAME']
        if 'SENTINEL' not in spacecraft.upper() and 'LANDSAT' not in spacecraft.upper():
            logger.info(f'ERROR the platform "{spacecraft}" is not supported')
            return False
    elif 'SENSOR' in meta:
        sensor = meta['SENSOR']
        pid = md.get('SENSOR_PRODUCT_ID', '')
        if 'OLI' in sensor and 'LC' in pid:
            i = pid.find('LC')
            spacecraft = f'Landsat-{int(pid[i + 2:i + 4])}'
        else:
            logger.info(f'ERROR the sensor "{sensor}" is not supported')
            return False
    else:
        logger.info('ERROR could not determine the platorm from metadata')
        return False
    md['SPACECRAFT_NAME'] = spacecraft
    if sensor is not None:
        names = [s.strip() for s in sensor.replace('_TIRS', '').split(';')]
        md['SENSOR'] = '; '.join(dict.fromkeys(names))
    elif 'SENTINEL' in spacecraft.upper():
        md['SENSOR'] = 'MSI'
    else:
        md['SENSOR'] = 'OLI'
    return True


def _load_hls_product_v2(file_list, image, md, flag_debug=False, alloc=None, engine=None):
    """Reads the seven band files of an HLS v2 product.  Returns False on failure.
    Fill detection and clipping are NOT done here: the raw planes and the fill values go
    to the kernel (A0 of the hot path).  With `engine` (pipeline.TileEngine) the planes come back RESIDENT in HBM
    (pipeline.DevicePlane): the host only inflates the blocks, the device undoes the predictor and untiles them."""
    logger.info('loading HLS v.2.0 layers:')
    image['fills'] = {}

    def find(key):
        landsat = 'SPACECRAFT_NAME' not in md or 'LANDSAT' in md['SPACECRAFT_NAME'].upper()
        suffix = (l30_v2_band_dict if landsat else s30_v2_band_dict)[key]
        return suffix, next((f for f in file_list if suffix + '.tif' in f), None)

    # an incomplete file list is reported before any raster is decoded
    def missing(band_dict):
        return [k for k, sfx in band_dict.items() if not any(sfx + '.tif' in f for f in file_list)]

    if missing(l30_v2_band_dict) and missing(s30_v2_band_dict):
        key = min(missing(l30_v2_band_dict), missing(s30_v2_band_dict), key=len)[0]
        logger.info(f'ERROR band {key} not found within list of input file(s)')
        return False
    def read(path):
        if engine is not None and not flag_debug:
            d = geotiff.open_geotiff(path)          # (an unreadable input is reported before the GPU is asked for)
            return (engine() if callable(engine) else engine).read_directory(d)
        return geotiff.read_geotiff(path, window=(0, 0, 1000, 1000) if flag_debug else None, alloc=alloc)

    # the first band's metadata decide the sensor, hence the file names of the other six, which
    # are then decoded side by side
    from concurrent.futures import ThreadPoolExecutor
    pending, pool = {}, ThreadPoolExecutor(6, thread_name_prefix='dswx-read')
    try:
        return _load_bands(file_list, image, md, flag_debug, find, read, pending, pool)
    finally:
        pool.shutdown(wait=True)


def _load_bands(file_list, image, md, flag_debug, find, read, pending, pool):
    keys = list(l30_v2_band_dict)
    for n_done, key in enumerate(keys):
        logger.info(f'    {key}')
        suffix, path = find(key)
        if path is None:
            logger.info(f'ERROR band {key} not found within list of input file(s)')
            return False
        try:
            arr, info = pending.pop(path).result() if path in pending else read(path)
        except (OSError, geotiff.GeoTiffError) as e:
            logger.info(f'ERROR could not open {path}: {e}')
            return False
        if n_done == 0:
            # harvested below for reflectance bands; needed now to name the remaining files
            if 'SPACECRAFT_NAME' not in md and not _harvest_hls_metadata(info.metadata, md):
                return False
            for other in keys[1:]:
                p = find(other)[1]
                if p is not None and p not in pending:
                    pending[p] = pool.submit(read, p)
        if flag_debug:
            logger.info('reading in debug mode')
        if 'hls_dataset_name' not in image:
            name = os.path.splitext(os.path.basename(path))[0]
            image['hls_dataset_name'] = name.replace(f'.{suffix}', '')
        fill = info.nodata
        if fill is None and '_FillValue' in info.metadata:
            fill = float(info.metadata['_FillValue'])
        elif fill is None:
            fill = -9999
        image['fills'][key] = fill
        image.setdefault('geo_tags', info.geo_tags)
        image.setdefault('geotransform', info.geotransform)
        image.setdefault('length', arr.shape[0])
        image.setdefault('width', arr.shape[1])
        if key == 'fmask':
            image[key] = arr if (isinstance(arr, pipeline.DevicePlane) and arr.dtype == np.uint8) else \
                np.ascontiguousarray(arr.numpy() if isinstance(arr, pipeline.DevicePlane) else arr, dtype=np.uint8)
            continue
        if arr.dtype != np.int16:
            logger.info(f'ERROR band {key} of {path} is {arr.dtype}, expected int16')
            return False
        if 'SPACECRAFT_NAME' not in md and not _harvest_hls_metadata(info.metadata, md):
            return False
        image[key] = arr
        image.setdefault('offset', {})[key] = float(info.metadata.get('add_offset', 0.0))
        image.setdefault('scale', {})[key] = float(info.metadata.get('scale_factor', 1.0))
    return True


# -----------------------------------------------------------------------------------
# metadata (:3817-4080)
# -----------------------------------------------------------------------------------
_LICENSE_TAIL = (' by law or by delegation do not assume any legal responsibility or'
                 ' liability, whether express or implied, arising from any use of this product.')


def _get_dswx_metadata_dict(product_id, product_version):
    md = OrderedDict()
    md['PRODUCT_ID'] = product_id
    md['PRODUCT_VERSION'] = product_version if product_version is not None else SOFTWARE_VERSION
    md['SOFTWARE_VERSION'] = SOFTWARE_VERSION
    md['PROJECT'] = 'OPERA'
    md['PRODUCT_LEVEL'] = '3'
    md['PRODUCT_TYPE'] = 'DSWx-HLS'
    md['PRODUCT_SOURCE'] = 'HLS'
    md['PROCESSING_DATETIME'] = datetime.now().strftime('%Y-%m-%dT%H:%M:%SZ')
    return md


def _source_field(description, path, missing):
    if description:
        return description
    if path:
        return os.path.basename(path)
    return missing


def _populate_dswx_metadata_datasets(md, hls_dataset, dem_file=None, dem_file_description=None,
                                     landcover_file=None, landcover_file_description=None,
                                     worldcover_file=None, worldcover_file_description=None,
                                     shoreline_shapefile=None,
                                     shoreline_shapefile_description=None):
    md['HLS_DATASET'] = hls_dataset
    md['DEM_SOURCE'] = _source_field(dem_file_description, dem_file, 'NOT_PROVIDED')
    text, copernicus = '', False
    if 'SENTINEL' in md['SPACECRAFT_NAME'].upper():
        copernicus = True
        text += ('This OPERA DSWx-HLS product contains modified Copernicus Sentinel Earth'
                 ' Observation (EO) data. Sentinel EO data is provided under COPERNICUS by the'
                 ' European Union and ESA; all rights reserved. Users, including those who'
                 ' redistribute, adapt, modify, or combine the contents of this product, must'
                 ' comply with the terms of the Copernicus Sentinel Data License Agreement. ')
    if 'COPERNICUS DEM' in md['DEM_SOURCE'].upper():
        copernicus = True
        text += ('This OPERA DSWx-HLS product contains modified Copernicus DEM data. The'
                 ' Copernicus DEM 30-m and Copernicus DEM 90-m were produced using Copernicus'
                 ' WorldDEM-30 © DLR e.V. 2010-2014 and © Airbus Defence and Space GmbH'
                 ' 2014-2018, provided under COPERNICUS by the European Union and ESA; all'
                 ' rights reserved. Users, including those who redistribute, adapt, modify, or'
                 ' combine the DEM layer (band 10) or derived SHAD layer (band 8), must comply'
                 ' with the terms of the Copernicus DEM License Agreement. For additional'
                 ' information, please refer to https://doi.org/10.5270/ESA-c5d3d65. ')
    who = ' in charge of the OPERA project and the Copernicus programme' if copernicus else \
        ' in charge of the OPERA project'
    md['LICENSE'] = text + 'The organizations' + who + _LICENSE_TAIL
    md['LANDCOVER_SOURCE'] = _source_field(landcover_file_description, landcover_file,
                                           'NOT_PROVIDED')
    md['WORLDCOVER_SOURCE'] = _source_field(worldcover_file_description, worldcover_file,
                                            'NOT_PROVIDED')
    md['SHORELINE_SOURCE'] = _source_field(shoreline_shapefile_description, shoreline_shapefile,
                                           'NOT_PROVIDED_OR_NOT_USED')


def _populate_dswx_metadata_processing_parameters(md, apply_ocean_masking,
                                                  apply_aerosol_class_remapping, aerosol_lists,
                                                  shadow_masking_algorithm, min_slope_angle,
                                                  max_sun_local_inc_angle,
                                                  mask_adjacent_to_cloud_mode,
                                                  forest_mask_landcover_classes,
                                                  ocean_masking_shoreline_distance_km):
    md['AEROSOL_CLASS_REMAPPING_ENABLED'] = 'TRUE' if apply_aerosol_class_remapping else 'FALSE'
    for key, values in zip(_AEROSOL_KEYS, aerosol_lists):
        # (sic) the reference keys this on forest_mask_landcover_classes, :4040-4045
        md[key.upper()] = ','.join(str(c) for c in values) \
            if forest_mask_landcover_classes else 'EMPTY'
    md['SHADOW_MASKING_ALGORITHM'] = shadow_masking_algorithm.upper()
    if shadow_masking_algorithm == 'sun_local_inc_angle':
        md['MIN_SLOPE_ANGLE'] = min_slope_angle
        md['MAX_SUN_LOCAL_INC_ANGLE'] = max_sun_local_inc_angle
    else:
        md['MIN_SLOPE_ANGLE'] = md['MAX_SUN_LOCAL_INC_ANGLE'] = 'NOT_USED'
    md['MASK_ADJACENT_TO_CLOUD_MODE'] = mask_adjacent_to_cloud_mode
    md['FOREST_MASK_LANDCOVER_CLASSES'] = \
        ','.join(str(c) for c in forest_mask_landcover_classes) \
        if forest_mask_landcover_classes else 'EMPTY'
    md['OCEAN_MASKING_ENABLED'] = 'TRUE' if apply_ocean_masking else 'FALSE'
    md['OCEAN_MASKING_SHORELINE_DISTANCE_KM'] = \
        ocean_masking_shoreline_distance_km if apply_ocean_masking else 'NOT_USED'


# -----------------------------------------------------------------------------------
# colour tables (:1381-1636, :2427-2575), as {value: (r, g, b)}
# -----------------------------------------------------------------------------------
def get_transparency_rgb_vals(top_rgb, bottom_rgb, alpha):
    if alpha < 0 or alpha > 1:
        raise ValueError('alpha must be in range [0, 1].')
    return tuple(int(alpha * a + (1 - alpha) * b) for a, b in zip(top_rgb, bottom_rgb))


def _get_interpreted_dswx_ctable(flag_collapse_wtr_classes=FLAG_COLLAPSE_WTR_CLASSES,
                                 layer_name='WTR'):
    ct = {0: (255, 255, 255)}
    if flag_collapse_wtr_classes:
        ct.update({1: (0, 0, 255), 2: (180, 213, 244)})
    else:
        ct.update({1: (0, 0, 255), 2: (95, 127, 255), 3: (0, 195, 0), 4: (150, 255, 150)})
    ct[WTR_OCEAN_MASKED] = OCEAN_MASKED_RGBA[:3]
    if layer_name == 'WTR':
        ct[WTR_CLOUD_MASKED] = (175, 175, 175)
        ct[WTR_SNOW_MASKED] = (0, 255, 255)
    ct[UINT8_FILL_VALUE] = FILL_VALUE_RGBA[:3]
    return ct


def _get_browse_ctable(flag_collapse_wtr_classes=FLAG_COLLAPSE_WTR_CLASSES, not_water_color='white',
                       cloud_color='gray', snow_color='cyan'):
    """Browse colour table (:1449-1526): WTR colours with the three display options."""
    if not_water_color not in ('white', 'nodata'):
        raise ValueError(f"not_water_color is {not_water_color}, but must be one of 'white' or 'nodata'")
    if cloud_color not in ('gray', 'nodata'):
        raise ValueError(f"cloud_color is {cloud_color}, but must be one of 'gray' or 'nodata'")
    if snow_color not in ('cyan', 'gray', 'nodata'):
        raise ValueError(f"snow_color is {snow_color}, but must be one of 'cyan', 'gray', or 'nodata'")
    ct = _get_interpreted_dswx_ctable(flag_collapse_wtr_classes=flag_collapse_wtr_classes)
    if snow_color == 'gray':
        ct[WTR_SNOW_MASKED] = ct[WTR_CLOUD_MASKED]
    elif snow_color == 'nodata':
        ct[WTR_SNOW_MASKED] = FILL_VALUE_RGBA[:3]
    ct[WTR_CLOUD_MASKED] = FILL_VALUE_RGBA[:3] if cloud_color == 'nodata' else (175, 175, 175)
    if not_water_color == 'nodata':
        ct[0] = FILL_VALUE_RGBA[:3]
    return ct


def geotiff2png(src_geotiff_filename, dest_png_filename, output_height=None, output_width=None,
                logger=None):
    """Paletted Byte GeoTIFF -> resized PNG, nearest neighbour (:2719-2783; integer layers
    only, which is all the workflow feeds it); the nodata index is transparent."""
    arr, info = geotiff.read_geotiff(src_geotiff_filename)
    if arr.ndim != 2 or arr.dtype != np.uint8:
        raise ValueError('geotiff2png handles single-band Byte rasters')
    h = arr.shape[0] if output_height is None else output_height
    w = arr.shape[1] if output_width is None else output_width
    small = geotiff.resample_nearest(arr, h, w)
    _makedirs(dest_png_filename)
    geotiff.write_png_palette(dest_png_filename, small, info.colormap,
                              transparent_index=None if info.nodata is None else int(info.nodata))
    (logger or logging.getLogger('proteus')).info(f'Browse Image PNG created: {dest_png_filename}')


def _save_output_rgb_file(red, green, blue, output_file, offset_dict, scale_dict,
                          flag_offset_and_scale_inputs, dswx_metadata_dict, geo_tags,
                          invalid_mask=None, output_files_list=None, flag_infrared=False):
    """Three-band Float32 reflectance composite (:2961-3054): scale * (float32(band) - offset),
    NaN on invalid pixels."""
    keys = ('swir1', 'nir', 'red') if flag_infrared else ('red', 'green', 'blue')
    planes = []
    import time as _time
    t_rgb = _time.perf_counter()
    for arr, key in zip((red, green, blue), keys):
        # the reference scales here unless the loader already did (flag_offset_and_scale_inputs, :3013): the same
        # statement on the same clipped planes either way; this host keeps the integer planes and scales here in both cases
        arr = scale_dict[key] * (np.asarray(arr, dtype=np.float32) - offset_dict[key])
        if invalid_mask is not None:
            arr[invalid_mask] = np.nan
        planes.append(np.asarray(arr, dtype=np.float32))
    _makedirs(output_file)
    stack = np.stack(planes)
    stages.add('host: RGB scaling', t_rgb, _time.perf_counter())

    def job():
        geotiff.write_geotiff(output_file, stack, geo_tags=geo_tags, metadata=dswx_metadata_dict,
                              overviews=geotiff.COG_OVERVIEW_FACTORS)
        logger.info(f'file saved: {output_file}')
    if output_files_list is not None:
        output_files_list.append(output_file)
    _run_or_defer(job)


def _save_output_rgb_planes(engine, planes, diag, scales, offsets, output_file, dswx_metadata_dict, geo_tags,
                            output_files_list=None):
    """_save_output_rgb_file (:2961-3054) for planes RESIDENT on the device: the scaling, the clip and the NaN mask are
    dswx_rgb_planes_device, the tiling and the floating-point predictor dswx_cog_blocks_device; DEFLATE on the host."""
    _makedirs(output_file)

    def job():
        levels = engine.rgb_levels(planes[0], planes[1], planes[2], diag, scales, offsets, FLAG_CLIP_NEGATIVE_REFLECTANCE,
                                   factors=geotiff.COG_OVERVIEW_FACTORS)
        geotiff.write_geotiff(output_file, None, levels=levels, geo_tags=geo_tags, metadata=dswx_metadata_dict)
        logger.info(f'file saved: {output_file}')
    if output_files_list is not None:
        output_files_list.append(output_file)
    _run_or_defer(job)


def _browse_png_job(browse, ctable, nodata, dest_png_filename, output_height, output_width):
    """What geotiff2png makes of the browse GeoTIFF (nearest-neighbour resize, palette, transparent nodata), from the
    plane itself."""
    def job():
        with stages.span('browse PNG (resample + encode)'):
            h = browse.shape[0] if output_height is None else output_height
            w = browse.shape[1] if output_width is None else output_width
            if isinstance(browse, pipeline.DevicePlane) and 0 < h <= 65535:
                small = browse.engine.resample_nearest(browse, h, w)            # gathered in HBM
            else:
                small = geotiff.resample_nearest(browse.numpy() if isinstance(browse, pipeline.DevicePlane) else np.asarray(browse), h, w)
            _makedirs(dest_png_filename)
            geotiff.write_png_palette(dest_png_filename, small, ctable,
                                      transparent_index=None if nodata is None else int(nodata))
        logger.info(f'Browse Image PNG created: {dest_png_filename}')
    return job


def _get_binary_water_ctable():
    return {0: (255, 255, 255), 1: (0, 0, 255), WTR_OCEAN_MASKED: OCEAN_MASKED_RGBA[:3],
            WTR_SNOW_MASKED: (0, 255, 255), WTR_CLOUD_MASKED: (175, 175, 175),
            UINT8_FILL_VALUE: FILL_VALUE_RGBA[:3]}


def _get_binary_mask_ctable():
    return {0: (64, 64, 64), 1: (255, 255, 255), WTR_OCEAN_MASKED: OCEAN_MASKED_RGBA[:3],
            UINT8_FILL_VALUE: FILL_VALUE_RGBA[:3]}


def _get_cloud_layer_ctable():
    base = [(255, 255, 255), (64, 64, 64), (0, 255, 255), (0, 127, 127), (192, 192, 192),
            (127, 127, 127), (255, 0, 255), (127, 127, 255)]
    ct = {i: c for i, c in enumerate(base)}
    ct[8] = (228, 205, 167)
    for i in range(9, 16):
        ct[i] = base[i - 8]
    ct[254] = OCEAN_MASKED_RGBA[:3]
    ct[UINT8_FILL_VALUE] = FILL_VALUE_RGBA[:3]
    return ct


def _get_confidence_layer_ctable():
    ct = _get_interpreted_dswx_ctable(flag_collapse_wtr_classes=False, layer_name='WTR')
    cloud_rgb, snow_rgb = ct[WTR_CLOUD_MASKED], ct[WTR_SNOW_MASKED]
    clear = [ct[c] for c in range(5)]
    ct[WTR_SNOW_MASKED] = ct[WTR_CLOUD_MASKED] = (0, 0, 0)
    for c in range(5):
        ct[10 + c] = get_transparency_rgb_vals(cloud_rgb, clear[c], 0.52)
        ct[20 + c] = snow_rgb
    return ct


# -----------------------------------------------------------------------------------
# writers (:2601-2716, :2786-2958) + save_as_cog (core.py:7-91): 512 x 512 tiled DEFLATE
# GeoTIFF in cloud-optimized layout; integer layers carry NEAREST overviews 4/16/64/128 as
# the reference builds them, Float32 files (RGB composites, DEM) cascaded CUBICSPLINE ones
# (core.py:41-46; GDAL's convolution restated in geotiff.py, on the device since round 6)
# -----------------------------------------------------------------------------------
def _makedirs(path):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)


class _DeferredWrites:
    """Collects the GeoTIFF writes of one product run and executes them side by side.  Every write
    already spreads its DEFLATE blocks over the codec's thread pool; running a few writes
    concurrently hides their serial parts (block assembly, overviews, file output).  File lists
    keep the order of the calls; the files exist once flush() / the `with` block returns.  The active
    collector is per THREAD: several product runs may be in flight in one process (proteus_amd.batch)."""
    _local = threading.local()

    def __init__(self, workers=4):
        self.jobs, self.workers = [], workers

    @classmethod
    def current(cls):
        return getattr(cls._local, 'active', None)

    @classmethod
    def reset(cls):
        cls._local.active = None

    def __enter__(self):
        _DeferredWrites._local.active = self
        return self

    def __exit__(self, exc_type, exc, tb):
        _DeferredWrites._local.active = None
        if exc_type is None:
            self.flush()

    def flush(self):
        jobs, self.jobs = self.jobs, []
        if len(jobs) < 2 or self.workers < 2:
            for job in jobs:
                job()
            return
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(self.workers, thread_name_prefix='dswx-write') as ex:
            list(ex.map(lambda job: job(), jobs))      # re-raises the first failure


def _run_or_defer(job):
    active = _DeferredWrites.current()
    if active is not None:
        active.jobs.append(job)
    else:
        job()


def _save_array(input_array, output_file, dswx_metadata_dict, geo_tags, description=None,
                output_files_list=None, ctable=None, no_data_value=None):
    _makedirs(output_file)
    resident = isinstance(input_array, pipeline.DevicePlane)
    integer = (input_array.dtype if resident else np.asarray(input_array).dtype).kind in 'uib'

    def job():
        if resident and (integer or input_array.dtype == np.float32):
            # the blocks of the image and of its overviews (NEAREST; Float32: CUBICSPLINE), predictor applied, made on the
            # device: the host only deflates (pipeline.TileEngine.layer_levels; save_as_cog, core.py:7-91)
            levels = input_array.engine.layer_levels(input_array, geotiff.COG_OVERVIEW_FACTORS)
            geotiff.write_geotiff(output_file, None, levels=levels, geo_tags=geo_tags, metadata=dswx_metadata_dict,
                                  nodata=no_data_value, descriptions=[description] if description else None,
                                  colormap=ctable)
            logger.info(f'file saved: {output_file}')
            return
        geotiff.write_geotiff(output_file, input_array.numpy() if resident else input_array, geo_tags=geo_tags,
                              metadata=dswx_metadata_dict, nodata=no_data_value,
                              descriptions=[description] if description else None, colormap=ctable,
                              overviews=geotiff.COG_OVERVIEW_FACTORS)
        logger.info(f'file saved: {output_file}')
    if output_files_list is not None:
        output_files_list.append(output_file)
    _run_or_defer(job)


def _gdal_byte(band_array):
    """What `gdal_band.WriteArray(array)` stores in a GDT_Byte band (GDALCopyWords): integers are clamped
    to 0..255; floating point is rounded half up after clamping, NaN becomes 0.  (GDAL is not installed
    here: the clamp / round rule is GDAL's documented conversion, unpinned by execution.)"""
    a = np.asarray(band_array)
    if a.dtype == np.uint8:
        return a
    if a.dtype == np.bool_:
        return a.astype(np.uint8)
    if a.dtype.kind in 'iu':
        return np.clip(a, 0, 255).astype(np.uint8)
    with np.errstate(invalid='ignore'):
        v = np.floor(np.clip(np.nan_to_num(a.astype(np.float64), nan=0.0), 0.0, 255.0) + 0.5)
    return np.minimum(v, 255.0).astype(np.uint8)


def save_dswx_product(layers, output_file, dswx_metadata_dict, geo_tags,
                      output_files_list=None):
    """Multi-band product (:2601-2707).  The file is created with the ten Byte bands of band_description_dict
    (:2663-2666), but the writing loop (:2673-2686) walks that dict and SKIPS, without advancing its band
    index, every name that was not passed to the call.  generate_dswx_layers passes wtr, bwtr, diag, wtr_1,
    wtr_2, land, shad, cloud, dem and never conf (:5383-5397), so a reference-made product is
        1 WTR, 2 BWTR, 3 DIAG, 4 WTR-1, 5 WTR-2, 6 LAND, 7 SHAD, 8 CLOUD, 9 DEM, 10 never written (zeros)
    and that is what `layers` (dict: name -> array or None; a name that is ABSENT is skipped the same way)
    produces here.  As there:
      * DIAG (UInt16 decimal digits) and DEM (Float32) go through GDAL's Byte conversion (`_gdal_byte`),
        i.e. they saturate at 255 (:2666 creates every band as GDT_Byte);
      * every written band carries the description of the FIRST band: `description` is assigned once from
        band_description_dict inside the loop and never reset (:2686-2687); unwritten bands have none;
      * WTR, WTR-1, WTR-2 arrive collapsed (the kernel applied _collapse_wtr_classes, :2688-2689).
    A layer that was passed as None (LAND / SHAD / DEM without their ancillary inputs; the reference
    would raise on WriteArray(None)) is a plane of nodata."""
    names = [n for n in band_description_dict if n in layers]
    first = next((np.asarray(layers[n]) for n in names if layers[n] is not None), None)
    if first is None:
        raise ValueError('save_dswx_product: no layer to save')
    shape = first.shape
    nbands = len(band_description_dict)
    stack = np.zeros((nbands,) + shape, dtype=np.uint8)             # an unwritten GTiff band reads as zeros
    for i, n in enumerate(names):
        stack[i] = UINT8_FILL_VALUE if layers[n] is None else _gdal_byte(layers[n])
    descriptions = [band_description_dict[names[0]]] * len(names) + [''] * (nbands - len(names))
    _makedirs(output_file)

    def job():
        geotiff.write_geotiff(output_file, stack, geo_tags=geo_tags, metadata=dswx_metadata_dict,
                              nodata=UINT8_FILL_VALUE, descriptions=descriptions,
                              overviews=geotiff.COG_OVERVIEW_FACTORS)
        logger.info(f'file saved: {output_file}')
    if output_files_list is not None:
        output_files_list.append(output_file)
    _run_or_defer(job)


def _as_plane(value, shape, name, dtype=np.uint8, engine=None):
    """Keyword-extension ancillary layer: ndarray, resident plane or GeoTIFF path on the HLS grid."""
    if value is None:
        return None
    if isinstance(value, (str, os.PathLike)) and engine is not None:
        value, _ = engine.read_plane(os.fspath(value))
    if isinstance(value, pipeline.DevicePlane):
        if value.shape != tuple(shape):
            raise ValueError(f'{name} has shape {value.shape}, the HLS grid is {tuple(shape)}')
        if value.dtype == np.dtype(dtype):
            return value
        value = value.numpy()
    if isinstance(value, (str, os.PathLike)):
        value, _ = geotiff.read_geotiff(os.fspath(value))
    arr = np.asarray(value)
    if arr.shape != tuple(shape):
        raise ValueError(f'{name} has shape {arr.shape}, the HLS grid is {tuple(shape)}')
    return np.ascontiguousarray(arr, dtype=dtype)


landcover_mask_type = 'standard'                                   # :41
landcover_threshold_dict = {'standard': [6, 3, 7, 3], 'water heavy': [6, 3, 7, 1]}   # :270-271
dswx_hls_landcover_classes_dict = {'low_intensity_developed_offset': 0,             # :252-264
                                   'high_intensity_developed_offset': 100,
                                   'water': 200, 'evergreen_forest': 201,
                                   'fill_value': UINT8_FILL_VALUE}


def _get_landcover_mask_ctable():
    """LAND layer colours (:1595-1636)."""
    ct = {dswx_hls_landcover_classes_dict['evergreen_forest']: (0, 255, 0),
          dswx_hls_landcover_classes_dict['water']: (0, 0, 255)}
    for i in range(100):
        ct[dswx_hls_landcover_classes_dict['low_intensity_developed_offset'] + i] = (255, 0, 255)
        ct[dswx_hls_landcover_classes_dict['high_intensity_developed_offset'] + i] = (255, 0, 0)
    ct[dswx_hls_landcover_classes_dict['fill_value']] = FILL_VALUE_RGBA[:3]
    return ct


def _grid_margin(info, geotransform, length, width, scale=1):
    """If the raster described by `info` is on the HLS grid refined `scale` times and covers it
    with the same margin on all four sides, returns that margin in (refined) pixels; else None.
    This is the test for "already warped": the reference gets there with gdal.Warp
    (`_warp` :4320-4420), which stays outside this package."""
    gt = info.geotransform
    if gt is None or gt[2] != 0 or gt[4] != 0:
        return None
    dx, dy = geotransform[1] / scale, geotransform[5] / scale
    if abs(gt[1] - dx) > 1e-9 * abs(dx) or abs(gt[5] - dy) > 1e-9 * abs(dy):
        return None
    mx, my = (geotransform[0] - gt[0]) / dx, (geotransform[3] - gt[3]) / dy
    m = round(mx)
    if m < 0 or abs(mx - m) > 1e-6 or abs(my - m) > 1e-6:
        return None
    if (info.height, info.width) != (scale * length + 2 * m, scale * width + 2 * m):
        return None
    return m


def _worldcover_year(metadata, worldcover_file_description):
    """Year of the ESA WorldCover map (:1060-1094): mid-point of time_start / time_end, else the
    first year 2000..2099 named in the file description, else 2000."""
    if 'time_start' in metadata and 'time_end' in metadata:
        fmt = '%Y-%m-%dT%H:%M:%SZ'
        t0, t1 = (datetime.strptime(metadata[k], fmt) for k in ('time_start', 'time_end'))
        year = (t0 + (t1 - t0) / 2.0).year
        logger.info(f'    ESA WorldCover map year: {year} (source: WorldCover file metadata)')
        return year
    logger.warning('WARNING Could not read the ESA WorldCover 10m metadata fields `time_start`'
                   ' and/or `time_end`')
    if worldcover_file_description:
        for year in range(2000, 2100):
            if str(year) in worldcover_file_description:
                logger.info(f'    ESA WorldCover map year: {year} (source: WorldCover file description)')
                return year
    logger.warning('WARNING Considering the ESA WorldCover 10m data year as 2000.')
    return 2000


def create_landcover_mask(copernicus_landcover_file, worldcover_file, worldcover_file_description,
                          output_file, scratch_dir, mask_type, geotransform, projection, length, width,
                          forest_mask_landcover_classes, dswx_metadata_dict=None,
                          output_files_list=None, temp_files_list=None, *, geo_tags=None, device=None, engine=None):
    """LAND layer from the Copernicus CGLS 100 m and ESA WorldCover 10 m maps (:906-1115) with the
    3 x 3 aggregation and class hierarchy on the GPU (dswx_landcover_mask_host).  The two
    reprojections of the reference (`_warp`, nearest, :970-992) are GDAL's: here both rasters must
    already be on the product grid -- CGLS on the HLS grid, WorldCover on the grid three times
    finer -- otherwise NotImplementedError.  Returns the uint8 mask (None if a file is missing)."""
    logger.info('creating LAND layer combining Copernicus Landcover 100m and ESA WorldCover 10m maps')
    for f in (copernicus_landcover_file, worldcover_file):
        if not os.path.isfile(f):
            logger.error(f'ERROR file not found: {f}')
            return None
    if engine is not None:
        # resident: both maps inflated on host threads, untiled on the device, aggregated there; the LAND layer stays in HBM
        # for the classifier and leaves as COG blocks (pipeline.TileEngine)
        cg_dir, wc_dir = geotiff.open_geotiff(copernicus_landcover_file), geotiff.open_geotiff(worldcover_file)
        cg_info, wc_info = cg_dir.info, wc_dir.info
    else:
        cg, cg_info = geotiff.read_geotiff(copernicus_landcover_file)
        wc, wc_info = geotiff.read_geotiff(worldcover_file)
    if _grid_margin(cg_info, geotransform, length, width) != 0 or \
            _grid_margin(wc_info, geotransform, length, width, scale=3) != 0:
        raise NotImplementedError(
            'landcover_file / worldcover_file are not on the product grid (CGLS: HLS grid; WorldCover: '
            '3x finer): reprojecting them needs GDAL, which stays on the host and is outside this '
            'drop-in (SURVEY.md section 2)')
    logger.info(f'    CGLS Land Cover 100m forest classes: {forest_mask_landcover_classes}')
    year = _worldcover_year(wc_info.metadata, worldcover_file_description)
    if engine is not None:
        planes = []
        for d in (wc_dir, cg_dir):
            plane, _ = engine.read_directory(d)
            if plane.dtype != np.uint8:
                plane = engine.upload(np.ascontiguousarray(plane.numpy(), dtype=np.uint8))
            planes.append(plane)
        land = engine.landcover_mask(planes[0], planes[1], forest_mask_landcover_classes,
                                     landcover_threshold_dict[mask_type.lower()], year - 2000)
    else:
        land = get_context(device).landcover_mask(
            wc, cg, forest_mask_landcover_classes, thresholds=landcover_threshold_dict[mask_type.lower()],
            year_offset=year - 2000)
    if output_file:
        _save_array(land, output_file, dswx_metadata_dict, geo_tags,
                    description=band_description_dict['LAND'], output_files_list=output_files_list,
                    ctable=_get_landcover_mask_ctable(), no_data_value=UINT8_FILL_VALUE)
    return land


# -----------------------------------------------------------------------------------
# the orchestrator (:4610-5417)
# -----------------------------------------------------------------------------------
def generate_dswx_layers(input_list,
                         output_file=None,
                         hls_thresholds=None,
                         dem_file=None,
                         dem_file_description=None,
                         output_interpreted_band=None,
                         output_rgb_file=None,
                         output_infrared_rgb_file=None,
                         output_binary_water=None,
                         output_confidence_layer=None,
                         output_diagnostic_layer=None,
                         output_non_masked_dswx=None,
                         output_shadow_masked_dswx=None,
                         output_landcover=None,
                         output_shadow_layer=None,
                         output_cloud_layer=None,
                         output_dem_layer=None,
                         output_browse_image=None,
                         browse_image_height=None,
                         browse_image_width=None,
                         exclude_psw_aggressive_in_browse=None,
                         not_water_in_browse=None,
                         cloud_in_browse=None,
                         snow_in_browse=None,
                         landcover_file=None,
                         landcover_file_description=None,
                         worldcover_file=None,
                         worldcover_file_description=None,
                         shoreline_shapefile=None,
                         shoreline_shapefile_description=None,
                         flag_offset_and_scale_inputs=False,
                         scratch_dir='.',
                         product_id=None,
                         product_version=SOFTWARE_VERSION,
                         check_ancillary_inputs_coverage=None,
                         apply_ocean_masking=None,
                         apply_aerosol_class_remapping=None,
                         aerosol_not_water_to_high_conf_water_fmask_values=None,
                         aerosol_water_moderate_conf_to_high_conf_water_fmask_values=None,
                         aerosol_partial_surface_water_conservative_to_high_conf_water_fmask_values=None,
                         aerosol_partial_surface_aggressive_to_high_conf_water_fmask_values=None,
                         shadow_masking_algorithm=None,
                         min_slope_angle=None,
                         max_sun_local_inc_angle=None,
                         mask_adjacent_to_cloud_mode=None,
                         forest_mask_landcover_classes=None,
                         ocean_masking_shoreline_distance_km=None,
                         flag_debug=False,
                         *,
                         landcover_mask=None,
                         shadow_layer=None,
                         ocean_mask=None,
                         device=None):
    """Compute the DSWx-HLS layers of one HLS tile; same signature and return value as the
    reference (:4610-4657) plus four keyword-only extensions (pre-gridded LAND / SHAD /
    ocean planes, and the GPU to use).  Returns True, or False after logging 'ERROR ...'
    when the input cannot be read (:4988-4990)."""
    _DeferredWrites.reset()                # a previous run that raised must not leave writes deferred
    local = locals()
    needs_defaults = [hls_thresholds, check_ancillary_inputs_coverage, apply_ocean_masking,
                      apply_aerosol_class_remapping, shadow_masking_algorithm, min_slope_angle,
                      max_sun_local_inc_angle, mask_adjacent_to_cloud_mode,
                      forest_mask_landcover_classes, ocean_masking_shoreline_distance_km,
                      browse_image_height, browse_image_width, exclude_psw_aggressive_in_browse,
                      not_water_in_browse, cloud_in_browse, snow_in_browse] + \
        [local[k] for k in _AEROSOL_KEYS]
    consts = parse_runconfig_file() if any(v is None for v in needs_defaults) else None

    def pick(value, name):
        return getattr(consts, name) if value is None else value


    if hls_thresholds is None:
        hls_thresholds = consts.hls_thresholds
    check_ancillary_inputs_coverage = pick(check_ancillary_inputs_coverage,
                                           'check_ancillary_inputs_coverage')
    apply_ocean_masking = pick(apply_ocean_masking, 'apply_ocean_masking')
    apply_aerosol_class_remapping = pick(apply_aerosol_class_remapping,
                                         'apply_aerosol_class_remapping')
    aerosol_lists = [pick(local[k], k) for k in _AEROSOL_KEYS]
    shadow_masking_algorithm = pick(shadow_masking_algorithm, 'shadow_masking_algorithm')
    min_slope_angle = pick(min_slope_angle, 'min_slope_angle')
    max_sun_local_inc_angle = pick(max_sun_local_inc_angle, 'max_sun_local_inc_angle')
    mask_adjacent_to_cloud_mode = pick(mask_adjacent_to_cloud_mode, 'mask_adjacent_to_cloud_mode')
    forest_mask_landcover_classes = pick(forest_mask_landcover_classes,
                                         'forest_mask_landcover_classes')
    ocean_masking_shoreline_distance_km = pick(ocean_masking_shoreline_distance_km,
                                               'ocean_masking_shoreline_distance_km')
    if scratch_dir is None:
        scratch_dir = '.'
    if product_id is None:
        product_id = os.path.splitext(os.path.basename(output_file))[0] if output_file \
            else 'dswx_hls'
    if isinstance(input_list, (str, os.PathLike)):
        input_list = [os.fspath(input_list)]

    logger.info(f'PROTEUS software version: {SOFTWARE_VERSION} (MI355X HIP per-pixel path)')
    logger.info('input files:')
    logger.info('    HLS product file(s):')
    for f in input_list:
        logger.info(f'        {f}')
    logger.info('product parameters:')
    logger.info(f'    product ID: {product_id}')
    logger.info(f'    product version: {product_version}')
    logger.info('processing parameters:')
    logger.info(f'    apply ocean masking: {apply_ocean_masking}')
    logger.info(f'    apply aerosol water class remapping: {apply_aerosol_class_remapping}')
    logger.info(f'    shadow masking algorithm: {shadow_masking_algorithm}')
    logger.info(f'    mask adjacent cloud/cloud-shadow mode: {mask_adjacent_to_cloud_mode}')
    if not apply_ocean_masking:
        shoreline_shapefile = shoreline_shapefile_description = None
        ocean_mask = None
    if shadow_masking_algorithm not in ('otsu', 'sun_local_inc_angle'):
        msg = f'ERROR Invalid shadow masking algorithm: {shadow_masking_algorithm}'
        logger.error(msg)
        raise ValueError(msg)
    if mask_adjacent_to_cloud_mode not in ('mask', 'ignore', 'cover'):
        msg = f'ERROR mask adjacent to cloud/cloud-shadow mode: {mask_adjacent_to_cloud_mode}'
        logger.info(msg)
        raise Exception(msg)
    if shoreline_shapefile is not None:
        if str(shoreline_shapefile).lower().endswith(('.tif', '.tiff')):
            # the same rule as for the DEM and the land-cover maps: an ancillary input that is ALREADY a raster
            # on the product grid is taken as it is -- here the ocean mask the reference would get by rasterising
            # the shoreline polygons (0 = ocean beyond the shoreline distance, :5243-5245)
            if ocean_mask is None:
                ocean_mask = os.fspath(shoreline_shapefile)
        else:
            raise NotImplementedError(
                'shoreline_shapefile: rasterising the shoreline needs GDAL/OGR, which stays on the host '
                'and is outside this drop-in (SURVEY.md section 2); pass the ocean mask already on the '
                'HLS grid with ocean_mask= (or give a GeoTIFF of it as shoreline_shapefile)')
    if dem_file is not None and shadow_masking_algorithm == 'otsu':
        raise NotImplementedError(
            "shadow_masking_algorithm 'otsu' thresholds GDAL's hillshade (gdal.DEMProcessing, "
            ":4160-4212), which stays on the host; use 'sun_local_inc_angle' (the default)")
    os.makedirs(scratch_dir, exist_ok=True)

    md = _get_dswx_metadata_dict(product_id, product_version)
    image = {}
    # HLS v1 (a single HDF4 file, :4972-4980) needs GDAL's HDF4 driver; every input goes
    # through the v2 per-band GeoTIFF loader, which reports what is missing
    # the blocks of the band files are inflated on host threads into page-locked memory; the device undoes the
    # predictor and untiles them: the planes are RESIDENT in HBM from here on (proteus_amd.pipeline)
    with stages.span('load HLS bands (7 files)'):
        ok = _load_hls_product_v2(list(input_list), image, md, flag_debug=flag_debug,
                                  alloc=lambda shape, dt: get_context(device).pinned_empty(shape, dt),
                                  engine=lambda: pipeline.engine_of(get_context(device)))
    if not ok:
        logger.info(f'ERROR could not read file(s): {input_list}')
        return False
    ctx = get_context(device)
    engine = pipeline.engine_of(ctx)
    for k in list(_capi.BAND_NAMES) + ['fmask']:          # (flag_debug reads a window on the host)
        if not isinstance(image[k], pipeline.DevicePlane):
            image[k] = engine.upload(image[k])
    version = '2.0'
    _populate_dswx_metadata_datasets(md, image['hls_dataset_name'], dem_file, dem_file_description,
                                     landcover_file, landcover_file_description,
                                     worldcover_file, worldcover_file_description,
                                     shoreline_shapefile, shoreline_shapefile_description)
    _populate_dswx_metadata_processing_parameters(
        md, apply_ocean_masking, apply_aerosol_class_remapping, aerosol_lists,
        shadow_masking_algorithm, min_slope_angle, max_sun_local_inc_angle,
        mask_adjacent_to_cloud_mode, forest_mask_landcover_classes,
        ocean_masking_shoreline_distance_km)
    logger.info(f'processing HLS {md["SPACECRAFT_NAME"][0]}30 dataset v.{version}')
    length, width = image['length'], image['width']
    geo_tags = image['geo_tags']

    # sun angles are required metadata (:5044-5059)
    def mean_angle(key):
        parts = md[key].split(', ')
        vals = [float(p) for p in parts]
        return sum(vals[:2]) / 2.0 if len(vals) == 2 else vals[0]

    sun_azimuth_angle = mean_angle('MEAN_SUN_AZIMUTH_ANGLE')
    sun_elevation_angle = 90 - mean_angle('MEAN_SUN_ZENITH_ANGLE')
    logger.info('Sun parameters (from HLS metadata):')
    logger.info(f'    mean azimuth angle: {sun_azimuth_angle}')
    logger.info(f'    mean elevation angle: {sun_elevation_angle}')

    shape = (length, width)
    dem = None
    early_list = []
    if dem_file is not None:
        # :5161-5186 with the warp taken out: the DEM must already be on the HLS grid with a
        # margin of DEM_MARGIN_IN_PIXELS (what `_warp(..., margin_in_pixels=50)` hands over).  Resident like the bands:
        # inflated on host threads, un-predicted (Float32 + PREDICTOR=3 as GDAL writes a DEM) and untiled on the device,
        # the SHAD layer and the cropped DEM layer made there (dswx_shadow_layer_device, dswx_copy_2d_device)
        logger.info(f'Preparing DEM file: {dem_file}')
        with stages.span('load DEM'):
            dem_dir = geotiff.open_geotiff(dem_file)
            dem_info = dem_dir.info
            margin = _grid_margin(dem_info, image['geotransform'], length, width) if dem_dir.spp == 1 else None
            if margin is None:
                raise NotImplementedError(
                    'dem_file is not on the HLS grid (same pixel size, same margin on all sides): '
                    'reprojecting it needs GDAL, which stays on the host and is outside this drop-in '
                    '(SURVEY.md section 2); warp it first or pass shadow_layer=')
            dem_with_margin, _ = engine.read_directory(dem_dir)
            if dem_with_margin.dtype != np.float32:                 # (the reference: np.asarray(..., dtype=np.float32))
                dem_with_margin = engine.upload(np.ascontiguousarray(dem_with_margin.numpy(), dtype=np.float32))
        if margin > DEM_MARGIN_IN_PIXELS:
            dem_with_margin = engine.crop(dem_with_margin, margin - DEM_MARGIN_IN_PIXELS)
            margin = DEM_MARGIN_IN_PIXELS
        elif margin < DEM_MARGIN_IN_PIXELS:
            logger.warning(f'WARNING DEM margin is {margin} pixels, the reference uses '
                           f'{DEM_MARGIN_IN_PIXELS}: slopes along the tile border differ')
            if margin < 2:
                logger.warning(f'WARNING terrain shadow layer with a margin of {margin} pixel(s): the general kernel '
                               '(dswx_shadow_v2) computes the whole raster, about three times slower than the filter kernel')
        sun, sin_az, cos_az, legacy = _shadow_geometry(sun_azimuth_angle, sun_elevation_angle)
        shadow_layer = engine.shadow_layer(dem_with_margin, sun, sin_az, cos_az, min_slope_angle, max_sun_local_inc_angle,
                                           margin, legacy)
        dem = engine.crop(dem_with_margin, margin)
        if output_dem_layer:
            _save_array(dem, output_dem_layer, md, geo_tags, description=band_description_dict['DEM'],
                        output_files_list=early_list, no_data_value=float('nan'))
    if landcover_file is not None and worldcover_file is not None:
        with stages.span('landcover mask (read + gpu)'):
            landcover_mask = create_landcover_mask(
                landcover_file, worldcover_file, worldcover_file_description, output_landcover, scratch_dir,
                landcover_mask_type, image['geotransform'], None, length, width,
                forest_mask_landcover_classes, dswx_metadata_dict=md, output_files_list=early_list,
                geo_tags=geo_tags, device=device, engine=engine)
        output_landcover = None          # saved by create_landcover_mask, as in the reference
    landcover_mask = _as_plane(landcover_mask, shape, 'landcover_mask', engine=engine)
    shadow_layer = _as_plane(shadow_layer, shape, 'shadow_layer', engine=engine)
    ocean_mask = _as_plane(ocean_mask, shape, 'ocean_mask', engine=engine)

    # ---- the hot path: one call into the HIP library ------------------------------
    bands = [image[k] for k in _capi.BAND_NAMES]
    params = _capi.make_params(
        hls_thresholds,
        band_fills=[image['fills'][k] for k in _capi.BAND_NAMES],
        fmask_fill=image['fills']['fmask'],
        clip_negative_reflectance=FLAG_CLIP_NEGATIVE_REFLECTANCE,
        mask_adjacent_to_cloud_mode=mask_adjacent_to_cloud_mode,
        apply_aerosol_class_remapping=apply_aerosol_class_remapping,
        aerosol_fmask_values=dict(zip((0, 2, 3, 4), aerosol_lists)),
        collapse_wtr_classes=FLAG_COLLAPSE_WTR_CLASSES,
        aerosol_max_nir=AEROSOL_REMAPPING_MAX_NIR,
        # flag_offset_and_scale_inputs (:2300-2302): the kernel scales the clipped reflectances to float32 with every
        # band's own scale_factor / add_offset metadata and runs the chain on those (generic kernel)
        offset_and_scale=[(image['scale'][k], image['offset'][k]) for k in _capi.BAND_NAMES]
        if flag_offset_and_scale_inputs else None,
        exclude_psw_aggressive_in_browse=pick(exclude_psw_aggressive_in_browse,
                                              'exclude_psw_aggressive_in_browse'),
        not_water_in_browse=pick(not_water_in_browse, 'not_water_in_browse'),
        cloud_in_browse=pick(cloud_in_browse, 'cloud_in_browse'),
        snow_in_browse=pick(snow_in_browse, 'snow_in_browse'),
        set_ocean_masked_to_nodata=True)
    wanted = ['diag', 'wtr1', 'wtr1_aerosol', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud']
    if output_browse_image:
        wanted.append('browse')
    import time as _time
    t_gpu = _time.perf_counter()
    res = engine.classify(bands, image['fmask'], params, land=landcover_mask, shad=shadow_layer, ocean=ocean_mask,
                          layers=tuple(wanted))
    t_gpu = _time.perf_counter() - t_gpu
    logger.info(f'    per-pixel chain on GPU: {engine.kernel_info} on resident planes')
    logger.info(f'    per-pixel chain incl. the upload of the ancillary planes: {t_gpu * 1e3:.1f} ms'
                f' ({length * width / t_gpu / 1e6:.0f} Mpixels/s)')
    n_valid, n_cloud_and_valid, n_not_ocean = (int(v) for v in res['counters'][0])

    # coverage metadata, floor percentages (:5113-5136)
    total = length * width
    logger.info('data coverage:')
    md['SPATIAL_COVERAGE'] = int(100 * float(n_valid) / total)
    md['SPATIAL_COVERAGE_EXCLUDING_MASKED_OCEAN'] = \
        0 if n_not_ocean == 0 else int(100 * float(n_valid) / n_not_ocean)
    md['CLOUD_COVERAGE'] = 0 if n_valid == 0 else int(100 * float(n_cloud_and_valid) / n_valid)
    logger.info(f'    spatial coverage [%]:  {md["SPATIAL_COVERAGE"]}')
    logger.info(f'    spatial coverage after ocean masking [%]:'
                f' {md["SPATIAL_COVERAGE_EXCLUDING_MASKED_OCEAN"]}')
    logger.info(f'    cloud coverage [%]:  {md["CLOUD_COVERAGE"]}')

    build_list, output_files_list = early_list, []
    collapse = FLAG_COLLAPSE_WTR_CLASSES
    writes = _DeferredWrites().__enter__()       # flushed before the browse PNG and at the end
    if shadow_layer is not None and output_shadow_layer:
        _save_array(shadow_layer, output_shadow_layer, md, geo_tags,
                    description=band_description_dict['SHAD'], output_files_list=build_list,
                    ctable=_get_binary_mask_ctable())
    if landcover_mask is not None and output_landcover:
        _save_array(landcover_mask, output_landcover, md, geo_tags,
                    description=band_description_dict['LAND'], output_files_list=build_list,
                    no_data_value=UINT8_FILL_VALUE)
    if output_diagnostic_layer:
        _save_array(res['diag'], output_diagnostic_layer, md, geo_tags,
                    description=band_description_dict['DIAG'], output_files_list=build_list,
                    no_data_value=DIAGNOSTIC_LAYER_NO_DATA_BINARY_REPR)
    for key, name, target in (('wtr1', 'WTR-1', output_non_masked_dswx),
                              ('wtr2', 'WTR-2', output_shadow_masked_dswx),
                              ('wtr', 'WTR', output_interpreted_band)):
        if target:
            _save_array(res[key], target, md, geo_tags, description=band_description_dict[name],
                        output_files_list=build_list, no_data_value=UINT8_FILL_VALUE,
                        ctable=_get_interpreted_dswx_ctable(collapse, layer_name=name))
    if output_cloud_layer:
        _save_array(res['cloud'], output_cloud_layer, md, geo_tags,
                    description=band_description_dict['CLOUD'], output_files_list=build_list,
                    no_data_value=UINT8_FILL_VALUE, ctable=_get_cloud_layer_ctable())
    if output_binary_water:
        _save_array(res['bwtr'], output_binary_water, md, geo_tags,
                    description=band_description_dict['BWTR'], output_files_list=build_list,
                    no_data_value=UINT8_FILL_VALUE, ctable=_get_binary_water_ctable())
    if output_confidence_layer:
        _save_array(res['conf'], output_confidence_layer, md, geo_tags,
                    description=band_description_dict['CONF'], output_files_list=build_list,
                    no_data_value=UINT8_FILL_VALUE, ctable=_get_confidence_layer_ctable())
    if output_rgb_file or output_infrared_rgb_file:
        # writer-side packaging (:5204-5223): the composites use the CLIPPED reflectances, scaled to float32, NaN on
        # invalid pixels -- all of it on the device from the resident planes (dswx_rgb_planes_device); the host deflates
        for target, keys in ((output_rgb_file, ('red', 'green', 'blue')),
                             (output_infrared_rgb_file, ('swir1', 'nir', 'red'))):
            if target:
                _save_output_rgb_planes(engine, [image[k] for k in keys], res['diag'],
                                        [image['scale'][k] for k in keys], [image['offset'][k] for k in keys],
                                        target, md, geo_tags, output_files_list)
    if output_browse_image:
        # browse = _compute_browse_array(WTR) from the kernel; full-res GeoTIFF + resized PNG
        # (:5301-5349)
        browse_tif = output_browse_image.replace('.png', '.tif')
        browse_ctable = _get_browse_ctable(collapse, pick(not_water_in_browse, 'not_water_in_browse'),
                                           pick(cloud_in_browse, 'cloud_in_browse'),
                                           pick(snow_in_browse, 'snow_in_browse'))
        _save_array(res['browse'], browse_tif, md, geo_tags, output_files_list=output_files_list,
                    no_data_value=UINT8_FILL_VALUE, ctable=browse_ctable)
        # the PNG is what geotiff2png renders from that GeoTIFF (:5335-5349) -- from the plane itself, without the read back
        browse_png_job = _browse_png_job(res['browse'], browse_ctable, UINT8_FILL_VALUE, output_browse_image,
                                         pick(browse_image_height, 'browse_image_height'),
                                         pick(browse_image_width, 'browse_image_width'))
        _run_or_defer(browse_png_job)
        output_files_list.append(output_browse_image)
    if output_file and not output_file.endswith('.vrt'):
        # the multi-band file carries the post-aerosol WTR-1 (in-place remap, :5260 -> :5389)
        # CONF is NOT passed (:5383-5397): the loop skips it and the bands after BWTR move up one (see there)
        _save_dswx_product_planes(engine, {'WTR': res['wtr'], 'BWTR': res['bwtr'], 'DIAG': res['diag'],
                                           'WTR-1': res['wtr1_aerosol'], 'WTR-2': res['wtr2'],
                                           'LAND': landcover_mask, 'SHAD': shadow_layer, 'CLOUD': res['cloud'], 'DEM': dem},
                                  output_file, md, geo_tags, output_files_list=output_files_list)
    elif output_file:
        logger.warning(f'VRT output "{output_file}" skipped: needs GDAL')
    writes.__exit__(None, None, None)
    logger.info('output files:')
    for f in build_list + output_files_list:
        logger.info(f'    {f}')
    return True


def _save_dswx_product_planes(engine, layers, output_file, dswx_metadata_dict, geo_tags, output_files_list=None):
    """save_dswx_product for layers RESIDENT on the device: the Byte conversion of DIAG / DEM (dswx_to_byte_device), the
    blocks, the NEAREST overviews and the predictor of all ten bands are made in HBM (pipeline.TileEngine.band_stack_levels);
    the band layout, the descriptions and the nodata planes are save_dswx_product's (see there)."""
    names = [n for n in band_description_dict if n in layers]
    first = next((layers[n] for n in names if layers[n] is not None), None)
    if first is None:
        raise ValueError('save_dswx_product: no layer to save')
    shape = tuple(first.shape)
    nbands = len(band_description_dict)
    descriptions = [band_description_dict[names[0]]] * len(names) + [''] * (nbands - len(names))
    _makedirs(output_file)

    def job():
        bands = []
        for n in names:
            a = layers[n]
            if a is None:
                bands.append(engine.constant_plane(shape, UINT8_FILL_VALUE))
            else:
                if not isinstance(a, pipeline.DevicePlane):
                    a = np.asarray(a)
                    if a.dtype.name not in ('uint8', 'bool', 'uint16', 'int16', 'float32'):
                        a = _gdal_byte(a)                           # a type the device conversion does not take
                    a = engine.upload(a)
                bands.append(engine.byte_plane(a))
        if nbands > len(names):
            bands += [engine.constant_plane(shape, 0)] * (nbands - len(names))    # an unwritten GTiff band reads as zeros
        levels = engine.band_stack_levels(bands, geotiff.COG_OVERVIEW_FACTORS)
        geotiff.write_geotiff(output_file, None, levels=levels, geo_tags=geo_tags, metadata=dswx_metadata_dict,
                              nodata=UINT8_FILL_VALUE, descriptions=descriptions)
        logger.info(f'file saved: {output_file}')
    if output_files_list is not None:
        output_files_list.append(output_file)
    _run_or_defer(job)


# -----------------------------------------------------------------------------------
# product comparison (:710-871)
# -----------------------------------------------------------------------------------
_METADATA_NOT_COMPARED = ('PROCESSING_DATETIME', 'DEM_SOURCE', 'LANDCOVER_SOURCE',
                          'WORLDCOVER_SOURCE', 'SOFTWARE_VERSION', 'SENSOR')


def _compare_dswx_hls_metadata(metadata_1, metadata_2):
    m1 = {k: v for k, v in metadata_1.items() if k != 'LICENSE'}
    m2 = {k: v for k, v in metadata_2.items() if k != 'LICENSE'}
    if len(m1) != len(m2):
        msg = (f'* input 1 metadata has {len(m1)} entries whereas input 2 metadata has'
               f' {len(m2)} entries.')
        if set(m1) - set(m2):
            msg += f' Input 1 metadata has extra entries with keys: {", ".join(set(m1) - set(m2))}.'
        if set(m2) - set(m1):
            msg += f' Input 2 metadata has extra entries with keys: {", ".join(set(m2) - set(m1))}.'
        return msg, False
    for k, v in m1.items():
        if k not in m2:
            return f'* the metadata key {k} is present in but it is not present in input 2', False
        if k in _METADATA_NOT_COMPARED:
            continue
        if m2[k] != v:
            return (f'* contents of metadata key {k} from input 1 has value "{v}" whereas the'
                    f' same key in input 2 metadata has value "{m2[k]}"'), False
    return None, True


def compare_dswx_hls_products(file_1, file_2):
    """Band-wise np.allclose(atol=1e-6, equal_nan=True), identical geotransform, identical
    metadata except LICENSE and the keys above (:710-784).  Prints an [OK]/[FAIL] report."""
    for f in (file_1, file_2):
        if not os.path.isfile(f):
            print(f'ERROR file not found: {f}')
            return False
    print('Comparing files:')
    print(f'    file 1: {file_1}')
    print(f'    file 2: {file_2}')
    a1, i1 = geotiff.read_geotiff(file_1)
    a2, i2 = geotiff.read_geotiff(file_2)
    all_ok = True

    def mark(flag):
        nonlocal all_ok
        all_ok = all_ok and flag
        return '[OK]   ' if flag else '[FAIL] '

    same = i1.bands == i2.bands
    print(f'{mark(same)}Comparing number of bands')
    if not same:
        print(' ' * 7 + f'Input 1 has {i1.bands} bands and input 2 has {i2.bands} bands')
        return False
    print('Comparing DSWx bands...')
    b1 = a1[None] if a1.ndim == 2 else a1
    b2 = a2[None] if a2.ndim == 2 else a2
    for b in range(i1.bands):
        same = b1[b].shape == b2[b].shape and bool(np.allclose(
            b1[b], b2[b], atol=COMPARE_DSWX_HLS_PRODUCTS_ERROR_TOLERANCE, equal_nan=True))
        print(f'{mark(same)}     Band {b + 1} - {i1.descriptions[b]}"')
        if not same and b1[b].shape == b2[b].shape:
            bad = np.argwhere(np.abs(b1[b].astype(np.float64) - b2[b].astype(np.float64)) >
                              COMPARE_DSWX_HLS_PRODUCTS_ERROR_TOLERANCE)
            if bad.size:
                y, x = bad[0]
                print(' ' * 7 + f'     * input 1 has value "{b1[b][y, x]}" in position (x: {x},'
                      f' y: {y}) whereas input 2 has value "{b2[b][y, x]}" in the same position.')
    same = bool(np.array_equal(i1.geotransform, i2.geotransform))
    print(f'{mark(same)}Comparing geotransform')
    if not same:
        print(' ' * 7 + f'* input 1 geotransform with content "{i1.geotransform}" differs from'
              f' input 2 geotransform with content "{i2.geotransform}".')
    msg, same = _compare_dswx_hls_metadata(i1.metadata, i2.metadata)
    print(f'{mark(same)}Comparing metadata')
    if not same:
        print(' ' * 7 + msg)
    return all_ok
