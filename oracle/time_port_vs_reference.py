"""TEST INFRASTRUCTURE -- build-container only (reads /root/reference, which does not exist on the GPU box).

Times the numpy port (oracle/dswx_oracle.py, what bench.py's `cpu_baseline` runs, kind "port") against
the reference's own functions, stage by stage, on one synthetic 3660 x 3660 tile, one thread.  Evidence
for how close the reported CPU baseline is to the real reference chain.

    python oracle/time_port_vs_reference.py > profiles/r02_cpu_port_vs_reference.json
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import _ref_import, dswx_oracle as o          # noqa: E402
from proteus_amd.synth import synth_tile                  # noqa: E402


def timed(f, *a, **k):
    t = time.perf_counter()
    r = f(*a, **k)
    return time.perf_counter() - t, r


def main():
    ref = _ref_import.import_reference()
    if ref is None:
        raise SystemExit('/root/reference is not present: run this in the build container')
    s = synth_tile(0, 3660, 3660)
    bands = [np.clip(b, 1, None) for b in s['bands']]
    fm = s['fmask']
    thr_ref = ref.HlsThresholds()
    for k, v in o.DEFAULT_THRESHOLDS.items():
        setattr(thr_ref, k, v)
    rows = {}

    def both(name, fr, fo):
        (t1, r1), (t2, r2) = fr(), fo()
        rows[name] = {'reference_s': round(t1, 3), 'port_s': round(t2, 3)}
        return r1, r2
    d1, d2 = both('_compute_diagnostic_tests', lambda: timed(ref._compute_diagnostic_tests, *bands, thr_ref),
                  lambda: timed(o.compute_diagnostic_tests, *bands, o.Thresholds()))
    w1, w2 = both('generate_interpreted_layer', lambda: timed(ref.generate_interpreted_layer, d1),
                  lambda: timed(o.generate_interpreted_layer, d2))
    both('_get_binary_representation', lambda: timed(ref._get_binary_representation, d1),
         lambda: timed(o.get_binary_representation, d2))
    c1, c2 = both('_compute_preliminary_cloud_layer', lambda: timed(ref._compute_preliminary_cloud_layer, fm, 'mask'),
                  lambda: timed(o.compute_preliminary_cloud_layer, fm, 'mask'))
    L = o.DEFAULT_AEROSOL_FMASK_VALUES
    both('_apply_aerosol_class_remapping',
         lambda: timed(ref._apply_aerosol_class_remapping, w1, bands[3], c1, fm, L[0], L[2], L[3], L[4]),
         lambda: timed(o.apply_aerosol_class_remapping, w2, bands[3], c2, fm))
    x1, x2 = both('_apply_landcover_and_shadow_masks',
                  lambda: timed(ref._apply_landcover_and_shadow_masks, w1, bands[3], None, None, thr_ref),
                  lambda: timed(o.apply_landcover_and_shadow_masks, w2, bands[3], None, None, o.Thresholds()))
    c1, c2 = both('_add_snow_to_cloud_layer', lambda: timed(ref._add_snow_to_cloud_layer, x1, c1, fm, 'mask'),
                  lambda: timed(o.add_snow_to_cloud_layer, x2, c2, fm, 'mask'))
    y1, y2 = both('_apply_cloud_masking', lambda: timed(ref._apply_cloud_masking, x1, c1),
                  lambda: timed(o.apply_cloud_masking, x2, c2))
    both('_get_binary_water_layer', lambda: timed(ref._get_binary_water_layer, y1),
         lambda: timed(o.get_binary_water_layer, y2))
    both('_get_confidence_layer', lambda: timed(ref._get_confidence_layer, x1, c1),
         lambda: timed(o.get_confidence_layer, x2, c2))
    both('_collapse_wtr_classes (one layer; the saved product pays it for WTR, WTR-1, WTR-2)',
         lambda: timed(ref._collapse_wtr_classes, y1), lambda: timed(o.collapse_wtr_classes, y2))
    assert np.array_equal(y1, y2) and np.array_equal(c1, c2) and np.array_equal(x1, x2)
    tot_r = sum(v['reference_s'] for v in rows.values())
    tot_p = sum(v['port_s'] for v in rows.values())
    print(json.dumps({'tile': [3660, 3660], 'numpy': np.__version__, 'cores': 1, 'host': 'build container',
                      'stages': rows, 'sum_reference_s': round(tot_r, 3), 'sum_port_s': round(tot_p, 3),
                      'port_over_reference': round(tot_p / tot_r, 3)}, indent=1))


if __name__ == '__main__':
    main()
