"""Helpers shared by the parity tests: load tests/golden/*.npz cases."""
import glob
import os
import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
THR_KEYS = ('wigt', 'awgt', 'pswt_1_mndwi', 'pswt_1_nir', 'pswt_1_swir1',
            'pswt_1_ndvi', 'pswt_2_mndwi', 'pswt_2_blue', 'pswt_2_nir',
            'pswt_2_swir1', 'pswt_2_swir2', 'lcmask_nir')
LAYERS = ('DIAG', 'WTR-1', 'WTR-1-AEROSOL', 'WTR-2', 'WTR', 'BWTR', 'CONF', 'CLOUD')
COLLAPSABLE = ('WTR', 'WTR-1', 'WTR-1-AEROSOL', 'WTR-2')


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def tile_case_names():
    return sorted(os.path.basename(p)[5:-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, 'tile_*.npz')))


def tile_case(name):
    z = load(f'tile_{name}.npz')
    case = {
        'bands': [np.ascontiguousarray(b) for b in z['in_bands']],
        'fmask': z['in_fmask'],
        'thr': dict(zip(THR_KEYS, z['thr'].tolist())),
        'band_fills': z['band_fills'].tolist(),
        'fmask_fill': float(z['fmask_fill']),
        'mode': str(z['mode']),
        'apply_aerosol': bool(z['apply_aerosol']),
        'aerosol_lists': {c: [int(v) for v in s.split(',')]
                          for c, s in zip((0, 2, 3, 4), z['aerosol_lists'].tolist())},
        'land': z['in_land'] if 'in_land' in z else None,
        'shad': z['in_shad'] if 'in_shad' in z else None,
        'ocean': z['in_ocean'] if 'in_ocean' in z else None,
        'offset_and_scale': [tuple(r) for r in z['offset_and_scale'].tolist()] if 'offset_and_scale' in z else None,
        'expected': {k[4:]: z[k] for k in z.files if k.startswith('out_')},
    }
    return case
