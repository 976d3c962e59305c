"""Runconfig loading and validation (host side, no GPU involved).

Mirrors `parse_runconfig_file` / `_deep_update` of PROTEUS
(src/proteus/dswx_hls.py:3575-3814) and the constraints of its yamale schema
(src/proteus/schemas/dswx_hls.yaml).  yamale and ruamel.yaml are not installed in
this image, so PyYAML loads the files and `validate_runconfig` implements the
schema's rules directly (types, enums, ranges, required keys, no unknown keys).
"""
import os

import yaml

DEFAULT_RUNCONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                 'defaults', 'dswx_hls.yaml')


class RunconfigError(ValueError):
    """Raised for schema violations (yamale raises YamaleError there, :3640)."""


def deep_update(main, update):
    """Nested dict merge; None in `update` never overwrites (:3575-3598)."""
    for key, val in update.items():
        if isinstance(val, dict):
            main[key] = deep_update(main.get(key) or {}, val)
        elif val is not None:
            main[key] = val
    return main


_SAFE_LOADER = getattr(yaml, 'CSafeLoader', yaml.SafeLoader)     # libyaml when PyYAML was built with it: the same documents, ~10 x faster


def load_yaml(path):
    with open(path) as fh:
        return yaml.load(fh, Loader=_SAFE_LOADER)


# ---- schema ------------------------------------------------------------------------
def _is_num(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _is_int(v):
    return isinstance(v, int) and not isinstance(v, bool)


def s_str(v):
    return isinstance(v, str)


def s_bool(v):
    return isinstance(v, bool)


def s_num(lo=None, hi=None):
    return lambda v: _is_num(v) and (lo is None or v >= lo) and (hi is None or v <= hi)


def s_int(lo=None):
    return lambda v: _is_int(v) and (lo is None or v >= lo)


def s_enum(*opts):
    return lambda v: v in opts


def s_list(item, min_len=0):
    return lambda v: isinstance(v, list) and len(v) >= min_len and all(item(x) for x in v)


REQ, OPT = True, False
_PROCESSING = {
    'check_ancillary_inputs_coverage': s_bool, 'apply_ocean_masking': s_bool,
    'apply_aerosol_class_remapping': s_bool,
    'aerosol_not_water_to_high_conf_water_fmask_values': s_list(_is_int),
    'aerosol_water_moderate_conf_to_high_conf_water_fmask_values': s_list(_is_int),
    'aerosol_partial_surface_water_conservative_to_high_conf_water_fmask_values':
        s_list(_is_int),
    'aerosol_partial_surface_aggressive_to_high_conf_water_fmask_values': s_list(_is_int),
    'shadow_masking_algorithm': s_enum('otsu', 'sun_local_inc_angle'),
    'min_slope_angle': s_num(-180, 180), 'max_sun_local_inc_angle': s_num(-180, 180),
    'mask_adjacent_to_cloud_mode': s_enum('mask', 'ignore', 'cover'),
    'forest_mask_landcover_classes': s_list(_is_int),
    'ocean_masking_shoreline_distance_km': s_num(),
}
for _l in ('wtr', 'bwtr', 'conf', 'diag', 'wtr_1', 'wtr_2', 'land', 'shad', 'cloud', 'dem',
           'rgb', 'infrared_rgb'):
    _PROCESSING['save_' + _l] = s_bool

SCHEMA = {
    'pge_name_group': {'pge_name': (REQ, s_enum('DSWX_HLS_PGE'))},
    'input_file_group': {'input_file_path': (REQ, s_list(s_str, 1))},
    'dynamic_ancillary_file_group': {k: (OPT, s_str) for k in (
        'dem_file', 'dem_file_description', 'landcover_file', 'landcover_file_description',
        'worldcover_file', 'worldcover_file_description', 'shoreline_shapefile',
        'shoreline_shapefile_description')},
    'primary_executable': {'product_type': (REQ, s_enum('DSWX_HLS'))},
    'product_path_group': {'product_path': (REQ, s_str), 'scratch_path': (REQ, s_str),
                           'output_dir': (REQ, s_str), 'product_id': (REQ, s_str),
                           'product_version': (OPT, s_num())},
    'processing': {k: (OPT, v) for k, v in _PROCESSING.items()},
    'browse_image_group': {
        'save_browse': (OPT, s_bool), 'browse_image_height': (OPT, s_int(1)),
        'browse_image_width': (OPT, s_int(1)),
        'exclude_psw_aggressive_in_browse': (OPT, s_bool),
        'not_water_in_browse': (OPT, s_enum('white', 'nodata')),
        'cloud_in_browse': (OPT, s_enum('gray', 'nodata')),
        'snow_in_browse': (OPT, s_enum('cyan', 'gray', 'nodata'))},
    'hls_thresholds': {k: (OPT, s_num()) for k in (
        'wigt', 'awgt', 'pswt_1_mndwi', 'pswt_1_nir', 'pswt_1_swir1', 'pswt_1_ndvi',
        'pswt_2_mndwi', 'pswt_2_blue', 'pswt_2_nir', 'pswt_2_swir1', 'pswt_2_swir2',
        'lcmask_nir')},
}
_REQUIRED_GROUPS = ('pge_name_group', 'input_file_group', 'dynamic_ancillary_file_group',
                    'primary_executable', 'product_path_group', 'processing',
                    'browse_image_group')


def validate_runconfig(doc, path='<runconfig>'):
    """Raises RunconfigError listing every violation of the schema."""
    errs = []
    rc = doc.get('runconfig') if isinstance(doc, dict) else None
    if not isinstance(rc, dict):
        raise RunconfigError(f'{path}: top-level key "runconfig" is missing')
    if not isinstance(rc.get('name'), str):
        errs.append('runconfig.name: required string')
    groups = rc.get('groups')
    if not isinstance(groups, dict):
        raise RunconfigError(f'{path}: runconfig.groups is missing')
    for g in _REQUIRED_GROUPS:
        if not isinstance(groups.get(g), dict):
            errs.append(f'runconfig.groups.{g}: required group is missing')
    for g, content in groups.items():
        if g not in SCHEMA:
            errs.append(f'runconfig.groups.{g}: unexpected group')
            continue
        if content is None:
            if g == 'hls_thresholds':
                continue
            content = {}
        for key, val in content.items():
            if key not in SCHEMA[g]:
                errs.append(f'runconfig.groups.{g}.{key}: unexpected key')
                continue
            required, check = SCHEMA[g][key]
            if val is None:
                if required:
                    errs.append(f'runconfig.groups.{g}.{key}: required value is missing')
                continue
            if not check(val):
                errs.append(f'runconfig.groups.{g}.{key}: invalid value {val!r}')
        for key, (required, _) in SCHEMA[g].items():
            if required and key not in content:
                errs.append(f'runconfig.groups.{g}.{key}: required key is missing')
    if errs:
        raise RunconfigError(f'Error validating {path}:\n\t' + '\n\t'.join(errs))
