#!/usr/bin/env python3
"""Wall time of one full product run (BASELINE.json configs[1] plumbing): a synthetic 3660 x 3660
HLS.L30 tile as seven DEFLATE GeoTIFFs -> bin/dswx_hls.py's generate_dswx_layers -> product layers,
with the GeoTIFF codec on 1 and on N threads.  Prints one JSON object."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_synthetic_hls as synth_hls          # noqa: E402
from proteus_amd import dswx_hls as D           # noqa: E402
from proteus_amd import stages                  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 3660
    out = {'size': size}
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        rcfile, files, _, _ = synth_hls.make(d, size=size)
        out['make_inputs_s'] = round(time.perf_counter() - t0, 2)
        parser = D.get_dswx_hls_cli_parser()
        for threads in ('1', '0'):
            os.environ['DSWX_IO_THREADS'] = threads
            args = parser.parse_args([rcfile])
            rc = D.parse_runconfig_file(user_runconfig_file=rcfile, args=args)
            best = None
            for rep in range(3):
                stages.start()
                t0 = time.perf_counter()
                ok = D.generate_dswx_layers(args.input_list, args.output_file, hls_thresholds=rc.hls_thresholds,
                                            product_id=args.product_id, product_version=args.product_version,
                                            scratch_dir=args.scratch_dir,
                                            output_interpreted_band=args.output_interpreted_band,
                                            output_binary_water=args.output_binary_water,
                                            output_confidence_layer=args.output_confidence_layer,
                                            output_diagnostic_layer=args.output_diagnostic_layer,
                                            output_non_masked_dswx=args.output_non_masked_dswx,
                                            output_shadow_masked_dswx=args.output_shadow_masked_dswx,
                                            output_cloud_layer=args.output_cloud_layer)
                dt = time.perf_counter() - t0
                rep_stages = stages.stop()
                assert ok
                if best is None or dt < best:
                    best, best_stages = dt, rep_stages
            tag = 'default' if threads == '0' else threads
            out[f'generate_dswx_layers_s_io_threads_{tag}'] = round(best, 3)
            out[f'stages_io_threads_{tag}'] = best_stages
        out['io_threads_default'] = min(32, os.cpu_count() or 1)
        out['kernel'] = D.get_context().last_kernel_info()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
