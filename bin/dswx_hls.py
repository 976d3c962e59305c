#!/usr/bin/env python3
"""`dswx_hls.py <runconfig.yaml | HLS files> [flags]` -- same entry point as PROTEUS
bin/dswx_hls.py:26-102, with the per-pixel chain on the MI355X."""
import logging
import mimetypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteus_amd.dswx_hls import (create_logger, generate_dswx_layers,   # noqa: E402
                                  get_dswx_hls_cli_parser, parse_runconfig_file,
                                  RunConfigConstants, _AEROSOL_KEYS)

logger = logging.getLogger('dswx_hls')


def main(argv=None):
    args = get_dswx_hls_cli_parser().parse_args(argv)
    create_logger(args.log_file, args.full_log_formatting)
    mimetypes.add_type('text/yaml', '.yaml', strict=True)
    mimetypes.add_type('text/yaml', '.yml', strict=True)
    guessed = mimetypes.guess_type(args.input_list[0])[0]
    first_is_text = guessed is not None and 'text' in guessed
    if len(args.input_list) > 1 and first_is_text:
        logger.info('ERROR only one runconfig file is allowed')
        return 1
    user_runconfig_file = args.input_list[0] if first_is_text else None
    consts = parse_runconfig_file(user_runconfig_file=user_runconfig_file, args=args)
    names = [f for f in RunConfigConstants._FIELDS]
    kwargs = {k: getattr(args, k) for k in names}
    for k in ('dem_file', 'dem_file_description', 'output_interpreted_band', 'output_rgb_file',
              'output_infrared_rgb_file', 'output_binary_water', 'output_confidence_layer',
              'output_diagnostic_layer', 'output_non_masked_dswx', 'output_shadow_masked_dswx',
              'output_landcover', 'output_shadow_layer', 'output_cloud_layer', 'output_dem_layer',
              'output_browse_image', 'landcover_file', 'landcover_file_description',
              'worldcover_file', 'worldcover_file_description', 'shoreline_shapefile',
              'shoreline_shapefile_description', 'flag_offset_and_scale_inputs', 'scratch_dir',
              'product_id', 'product_version', 'flag_debug', 'landcover_mask', 'shadow_layer',
              'ocean_mask', 'device'):
        kwargs[k] = getattr(args, k)
    ok = generate_dswx_layers(args.input_list, args.output_file,
                              hls_thresholds=consts.hls_thresholds, **kwargs)
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
