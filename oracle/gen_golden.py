#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates tests/golden/*.npz from the REAL reference.

Run in the build container only (needs /root/reference):

    python oracle/gen_golden.py

It imports the reference's `proteus.dswx_hls` (oracle/_ref_import.py), feeds its
own per-pixel functions with exhaustive tables, known-answer vectors and seeded
synthetic tiles, and stores inputs + the reference's outputs as small fixtures.
The fixtures are data (inputs and expected outputs); no reference source travels.
The tile cases replay the reference orchestrator's hot segment
(src/proteus/dswx_hls.py:5088-5112, :5225-5286, :5358-5369) call by call, because
`generate_dswx_layers` itself needs GDAL, which is not installed here.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle._ref_import import import_reference  # noqa: E402
from proteus_amd.synth import synth_tile          # noqa: E402

# DSWX_GOLDEN_OUT: write somewhere else (tests/test_oracle_golden.py::test_goldens_reproduce_from_reference regenerates
# into a temporary directory and compares with the committed fixtures)
GOLDEN = os.environ.get('DSWX_GOLDEN_OUT') or os.path.join(ROOT, 'tests', 'golden')

ALT_THRESHOLDS = {
    'default': {},
    'fractional': dict(wigt=0.1, awgt=-12.25, pswt_1_mndwi=-0.3, pswt_1_nir=1499.5,
                       pswt_1_swir1=900.25, pswt_1_ndvi=0.55, pswt_2_mndwi=-0.25,
                       pswt_2_blue=999.9, pswt_2_nir=2500.5, pswt_2_swir1=3000.75,
                       pswt_2_swir2=1000.125, lcmask_nir=1199.5),
    'zeros': dict(wigt=0.0, awgt=0.3, pswt_1_mndwi=0.0, pswt_1_ndvi=0.0,
                  pswt_2_mndwi=-1.0),
    'thirds': dict(wigt=1.0 / 3.0, pswt_1_mndwi=-1.0 / 3.0, pswt_1_ndvi=2.0 / 3.0,
                   pswt_2_mndwi=-0.2),
    # the band thresholds of the defaults in reflectance units (x 1e-4): what a caller of
    # flag_offset_and_scale_inputs has to supply for the tests to mean anything
    'reflectance': dict(pswt_1_nir=0.15, pswt_1_swir1=0.09, pswt_2_blue=0.1, pswt_2_nir=0.25,
                        pswt_2_swir1=0.3, pswt_2_swir2=0.1, lcmask_nir=0.12),
}


def make_thresholds(ref, **kw):
    thr = ref.HlsThresholds()
    vals = dict(wigt=0.124, awgt=0.0, pswt_1_mndwi=-0.44, pswt_1_nir=1500,
                pswt_1_swir1=900, pswt_1_ndvi=0.7, pswt_2_mndwi=-0.5,
                pswt_2_blue=1000, pswt_2_nir=2500, pswt_2_swir1=3000,
                pswt_2_swir2=1000, lcmask_nir=1200)
    vals.update(kw)
    for k, v in vals.items():
        setattr(thr, k, v)
    return thr


DEFAULT_AEROSOL = ([224, 160, 96], [224, 160, 96],
                   [224, 192, 160, 128, 96], [224, 192, 160, 128, 96])
CUSTOM_AEROSOL = ([224, 226, 2, 12, 96], [160, 164], [192, 200, 72], [128, 130, 255, 0])


# ---------------------------------------------------------------------------
def gen_tables(ref):
    out = {}
    # A6: the reference's own unit-test vector (tests/test_dswx_hls_units.py:7-28)
    keys = list(ref.interpreted_dswx_band_dict.keys())
    unit_in = np.full((1, len(keys) + 1), 111111)
    for i, k in enumerate(keys):
        unit_in[0, i] = k
    out['interp_unit_in'] = unit_in
    out['interp_unit_out'] = ref.generate_interpreted_layer(unit_in)
    d = np.arange(64, dtype=np.uint16).reshape(1, -1)
    out['interp_in'] = d
    out['interp_out'] = ref.generate_interpreted_layer(d)
    # A7
    out['binrepr_in'] = d
    out['binrepr_out'] = ref._get_binary_representation(d.copy())
    # A15
    v = np.arange(256, dtype=np.uint8).reshape(1, -1)
    out['collapse_in'] = v
    out['collapse_out'] = ref._collapse_wtr_classes(v)
    # A13
    out['bwtr_in'] = v
    out['bwtr_out'] = ref._get_binary_water_layer(v)
    # A2
    for mode in ('mask', 'ignore', 'cover'):
        out['prelim_' + mode] = ref._compute_preliminary_cloud_layer(v, mode)
    # A9 on a full grid, default and custom lists
    fm, cls, nr, cl = np.meshgrid(
        np.arange(256, dtype=np.uint8),
        np.array([0, 1, 2, 3, 4, 254, 255], dtype=np.uint8),
        np.array([1, 1000, 1001, -3], dtype=np.int16),
        np.array([0, 1, 4, 5, 255], dtype=np.uint8), indexing='ij')
    out['aer_fmask'], out['aer_cls'], out['aer_nir'], out['aer_cloud'] = \
        fm, cls, nr, cl
    for tag, lists in (('default', DEFAULT_AEROSOL), ('custom', CUSTOM_AEROSOL)):
        w = cls.copy()
        c = cl.copy()
        ref._apply_aerosol_class_remapping(w, nr, c, fm, *lists)
        out[f'aer_{tag}_wtr1'] = w
        out[f'aer_{tag}_cloud'] = c
    # A10
    ld, sh, cls, nr = np.meshgrid(
        np.arange(256, dtype=np.uint8), np.array([0, 1], dtype=np.uint8),
        np.array([0, 1, 2, 3, 4, 254, 255], dtype=np.uint8),
        np.array([1200, 1201, 1], dtype=np.int16), indexing='ij')
    out['lc_land'], out['lc_shad'], out['lc_cls'], out['lc_nir'] = ld, sh, cls, nr
    thr = make_thresholds(ref)
    out['lc_both'] = ref._apply_landcover_and_shadow_masks(cls, nr, ld, sh.astype(bool), thr)
    out['lc_both_u8shad'] = ref._apply_landcover_and_shadow_masks(cls, nr, ld, sh, thr)
    out['lc_land_only'] = ref._apply_landcover_and_shadow_masks(cls, nr, ld, None, thr)
    out['lc_shad_only'] = ref._apply_landcover_and_shadow_masks(cls, nr, None, sh.astype(bool), thr)
    out['lc_none'] = ref._apply_landcover_and_shadow_masks(cls, nr, None, None, thr)
    # A11 (non-cover), A12, A14 on grids
    fm, cl, w2 = np.meshgrid(
        np.arange(256, dtype=np.uint8),
        np.array([0, 1, 4, 5, 8, 9, 12, 13], dtype=np.uint8),
        np.array([0, 1, 2, 3, 4, 254, 255], dtype=np.uint8), indexing='ij')
    out['snow_fmask'], out['snow_cloud_in'], out['snow_wtr2'] = fm, cl, w2
    out['snow_cloud_out'] = ref._add_snow_to_cloud_layer(w2, cl.copy(), fm, 'mask')
    w2, cl = np.meshgrid(np.arange(256, dtype=np.uint8),
                         np.arange(256, dtype=np.uint8), indexing='ij')
    out['cm_wtr2'], out['cm_cloud'] = w2, cl
    out['cm_wtr'] = ref._apply_cloud_masking(w2, cl)
    out['cm_conf'] = ref._get_confidence_layer(w2, cl)
    np.savez_compressed(os.path.join(GOLDEN, 'tables.npz'), **out)
    print('tables.npz', len(out), 'arrays')


# ---------------------------------------------------------------------------
def tie_vectors():
    """Band 6-tuples that sit exactly on, and one count either side of, every
    rational value the quotient thresholds can hit, plus the survey's KATs."""
    rows = [
        (300, 400, 300, 200, 100, 50), (500, 600, 700, 3000, 2500, 1500),
        (100, 281, 300, 1700, 219, 50), (100, 282, 300, 1699, 219, 50),
        (100, 100, 100, 100, 300, 100), (100, 101, 100, 100, 300, 100),
        (100, 700, 100, 100, 1800, 100), (1, 1, 1, 1, 1, 1),
        (20000,) * 6, (999, 3000, 2000, 2499, 2999, 999),
        (1000, 3000, 2000, 2500, 3000, 1000), (400, 1000, 600, 1499, 899, 300),
        (1, 5000, 1, 1, 1, 32767), (32767,) * 6, (32767, 1, 32767, 1, 32767, 1),
        (1, 32767, 1, 32767, 1, 32767), (16384, 16384, 16384, 16384, 16384, 16384),
        (16383, 16384, 16385, 16383, 16384, 16385),
    ]
    # mndwi = (g-s1)/(g+s1) == p/q  <=>  g = (q+p)k/2, s1 = (q-p)k/2
    for p, q in ((31, 250), (-11, 25), (-1, 2), (1, 10), (-3, 10), (-1, 4),
                 (0, 1), (1, 3), (-1, 3), (-1, 5), (-1, 1)):
        for k in (2, 4, 10, 22, 50, 64, 100, 130, 200):
            g2, s2 = (q + p) * k, (q - p) * k
            if g2 % 2 or s2 % 2:
                continue
            g, s1 = g2 // 2, s2 // 2
            for dg in (-1, 0, 1):
                for ds in (-1, 0, 1):
                    gg, ss = g + dg, s1 + ds
                    if 1 <= gg <= 32767 and 1 <= ss <= 32767:
                        rows.append((100, gg, 100, 100, ss, 100))
                        rows.append((900, gg, 300, 1400, ss, 900))
    # ndvi = (n-r)/(n+r) == p/q
    for p, q in ((7, 10), (11, 20), (0, 1), (2, 3), (1, 2)):
        for k in (2, 4, 10, 20, 60, 100, 170, 400):
            n2, r2 = (q + p) * k, (q - p) * k
            if n2 % 2 or r2 % 2:
                continue
            n, r = n2 // 2, r2 // 2
            for dn in (-1, 0, 1):
                for dr in (-1, 0, 1):
                    nn, rr = n + dn, r + dr
                    if 1 <= nn <= 32767 and 1 <= rr <= 32767:
                        rows.append((100, 500, rr, nn, 300, 100))
                        rows.append((100, 500, rr, nn, 899, 100))
    return np.asarray(rows, dtype=np.int16)


def gen_diag(ref):
    rng = np.random.default_rng(20251010)
    parts = [tie_vectors()]
    # full positive int16 range, uniformly random (lots of int16 wrap)
    parts.append(rng.integers(1, 32768, size=(60000, 6)).astype(np.int16))
    # dense cloud around the integer thresholds
    base = np.array([1000, 1100, 900, 1500, 900, 1000])
    parts.append((base + rng.integers(-40, 41, size=(60000, 6))).astype(np.int16))
    # values incl. negatives / zero (clip disabled at this level: the function
    # sees whatever the caller passes) -> exercises n/0 and 0/0
    parts.append(rng.integers(-300, 300, size=(30000, 6)).astype(np.int16))
    special = np.array([(5, 7, 3, 0, -7, 2), (5, -7, 3, 0, 7, 2), (0, 0, 0, 0, 0, 0),
                        (9, 4, 9, -9, -4, 1), (9, 4, -9, 9, -4, 1),
                        (1, -32768, 1, 1, -32768, 1), (1, -32768, 1, 1, 32767, 1)],
                       dtype=np.int16)
    parts.append(special)
    vec = np.concatenate(parts, axis=0)
    out = {'bands': vec}
    cols = [np.ascontiguousarray(vec[:, i]).reshape(1, -1) for i in range(6)]
    with np.errstate(all='ignore'):
        for tag, kw in ALT_THRESHOLDS.items():
            thr = make_thresholds(ref, **kw)
            out['diag_' + tag] = ref._compute_diagnostic_tests(*cols, thr)
            out['thr_' + tag] = np.array(
                [getattr(thr, k) for k in THR_KEYS], dtype=np.float64)
        # float indices, for the 1e-6 debug planes
        g, s1, n, r = cols[1], cols[4], cols[3], cols[2]
        out['mndwi'] = (g - s1) / (g + s1)
        out['ndvi'] = (n - r) / (n + r)
        out['awesh'] = cols[0] + (2.5 * g) - (1.5 * (n + s1)) - (0.25 * cols[5])
    np.savez_compressed(os.path.join(GOLDEN, 'diag_vectors.npz'), **out)
    print('diag_vectors.npz', vec.shape[0], 'vectors')


THR_KEYS = ('wigt', 'awgt', 'pswt_1_mndwi', 'pswt_1_nir', 'pswt_1_swir1',
            'pswt_1_ndvi', 'pswt_2_mndwi', 'pswt_2_blue', 'pswt_2_nir',
            'pswt_2_swir1', 'pswt_2_swir2', 'lcmask_nir')


# ---------------------------------------------------------------------------
def run_reference_chain(ref, bands_raw, fmask, thr, land, shad, ocean, mode,
                        apply_aerosol, aerosol_lists, band_fills, fmask_fill, offset_and_scale=None):
    """The reference's own functions in the orchestrator's order."""
    # A0 -- _load_hls_band_from_file :2195-2209, :2298-2299 (GDAL part skipped)
    invalid = None
    for img, fill in list(zip(bands_raw, band_fills)) + [(fmask, fmask_fill)]:
        eq = img == fill
        invalid = eq if invalid is None else np.logical_or(invalid, eq)
    assert ref.FLAG_CLIP_NEGATIVE_REFLECTANCE
    clipped = [np.clip(b, 1, None) for b in bands_raw]
    if offset_and_scale is not None:
        # flag_offset_and_scale_inputs, the reference's own statement (:2300-2302) with its float() metadata values (:2295-2298)
        clipped = [float(scale_factor) * (np.asarray(image, dtype=np.float32) - float(offset))
                   for image, (scale_factor, offset) in zip(clipped, offset_and_scale)]
    blue, green, red, nir, swir1, swir2 = clipped
    invalid_ind = np.where(invalid)
    valid_array = ~invalid
    # :5088-5136
    cloud = ref._compute_preliminary_cloud_layer(fmask, mode)
    total = fmask.size
    if ocean is not None:
        valid_array = np.logical_and(valid_array, ocean)
        n_not_ocean = np.sum(ocean)
    else:
        n_not_ocean = total
    n_valid = np.sum(valid_array)
    n_cloud_and_valid = np.sum((cloud != 0) & valid_array)
    spatial = int(100 * float(n_valid) / total)
    cloud_cov = 0 if n_valid == 0 else int(100 * float(n_cloud_and_valid) / n_valid)
    spatial_no = 0 if n_not_ocean == 0 else int(100 * float(n_valid) / n_not_ocean)
    # :5225-5249
    with np.errstate(all='ignore'):
        dd = ref._compute_diagnostic_tests(blue, green, red, nir, swir1, swir2, thr)
    dd[invalid_ind] = ref.DIAGNOSTIC_LAYER_NO_DATA_DECIMAL
    wtr_1 = ref.generate_interpreted_layer(dd)
    diag = ref._get_binary_representation(dd)
    if ocean is not None:
        wtr_1[ocean == 0] = ref.WTR_OCEAN_MASKED
    wtr_1[invalid_ind] = ref.UINT8_FILL_VALUE
    wtr_1_saved = wtr_1.copy()
    # :5260-5286
    if apply_aerosol:
        ref._apply_aerosol_class_remapping(wtr_1, nir, cloud, fmask, *aerosol_lists)
    wtr_2 = ref._apply_landcover_and_shadow_masks(wtr_1, nir, land, shad, thr)
    cloud = ref._add_snow_to_cloud_layer(wtr_2, cloud, fmask, mode)
    wtr = ref._apply_cloud_masking(wtr_2, cloud)
    bwtr = ref._get_binary_water_layer(wtr)
    conf = ref._get_confidence_layer(wtr_2_layer=wtr_2, cloud_layer=cloud)
    res = {'DIAG': diag, 'WTR-1': wtr_1_saved, 'WTR-1-AEROSOL': wtr_1,
           'WTR-2': wtr_2, 'WTR': wtr, 'BWTR': bwtr, 'CONF': conf, 'CLOUD': cloud}
    for name in ('WTR', 'WTR-1', 'WTR-1-AEROSOL', 'WTR-2'):
        res[name + '.collapsed'] = ref._collapse_wtr_classes(res[name])
    res['counters'] = np.array([n_valid, n_cloud_and_valid, n_not_ocean,
                                spatial, cloud_cov, spatial_no], dtype=np.int64)
    return res


TILE_CASES = [
    # name, tile, H, W, masks(land, shad, ocean), mode, aerosol, lists, thresholds, fills
    dict(name='t64_plain', tile=0, H=64, W=64),
    dict(name='t128_all_masks', tile=1, H=128, W=128, land=1, shad=1, ocean=1),
    dict(name='t100x37_ragged', tile=2, H=100, W=37, land=1, shad=1, ocean=1),
    dict(name='t96_land_only', tile=3, H=96, W=96, land=1),
    dict(name='t96_shad_only', tile=4, H=96, W=96, shad=1),
    dict(name='t96_ocean_only', tile=5, H=96, W=96, ocean=1),
    dict(name='t128_ignore', tile=6, H=128, W=128, land=1, shad=1, mode='ignore'),
    dict(name='t128_no_aerosol', tile=7, H=128, W=128, land=1, shad=1, ocean=1,
         aerosol=False),
    dict(name='t128_custom_aerosol', tile=8, H=128, W=128, land=1, shad=1,
         lists='custom'),
    dict(name='t128_fractional_thr', tile=9, H=128, W=128, land=1, shad=1, ocean=1,
         thr='fractional'),
    dict(name='t128_thirds_thr', tile=10, H=128, W=128, thr='thirds'),
    dict(name='t128_zeros_thr', tile=11, H=128, W=128, thr='zeros'),
    dict(name='t64_other_fills', tile=12, H=64, W=64, fills='other'),
    dict(name='t256_all_masks', tile=13, H=256, W=256, land=1, shad=1, ocean=1),
    dict(name='t1x1', tile=14, H=1, W=1),
    dict(name='t3x5', tile=15, H=3, W=5, land=1, shad=1, ocean=1),
    dict(name='t17x16', tile=16, H=17, W=16, land=1, shad=1, ocean=1),
    dict(name='t160_cover', tile=17, H=160, W=160, land=1, shad=1, ocean=1,
         mode='cover'),
    # round 2: more of the reference's own 'cover' outputs -- rasters wider / taller than one 222-pixel window
    # of the dilation kernel in both directions, large coherent snow fields that the 10 + 7 steps really travel
    # through, other plane sets / aerosol settings
    dict(name='t300x470_cover_plain', tile=18, H=300, W=470, mode='cover', blobs='discs'),
    dict(name='t250x230_cover_land_custom', tile=19, H=250, W=230, land=1, mode='cover', blobs='discs',
         lists='custom'),
    dict(name='t96x120_cover_no_aerosol', tile=20, H=96, W=120, shad=1, ocean=1, mode='cover', aerosol=False),
    # round 3: the ITERATION COUNTS of the two masked dilations (:2060 iterations=10, :2075 iterations=7) pinned on the
    # reference's own output: one-pixel-wide corridors of adjacent-to-cloud pixels of length 1 ... 25 with a snow seed
    # at one end / in the middle / outside, over land and over water (where the second dilation grows back), cloud
    # pixels blocking the way, corridors that run into the raster edge or across the 222-pixel window seams of the
    # dilation kernel in x and in y, two-pixel-wide and L-shaped ones
    dict(name='t260x300_cover_corridors', tile=21, H=260, W=300, mode='cover', blobs='corridors'),
    # round 3: flag_offset_and_scale_inputs -- the chain on float32 reflectances (:2300-2302), with thresholds in
    # reflectance units, with the default (digital-number) thresholds, and with a different scale / offset per band
    dict(name='t96_scaled_reflectance_thr', tile=22, H=96, W=96, land=1, shad=1, ocean=1, thr='reflectance',
         scale=[(0.0001, 0.0)] * 6),
    dict(name='t64_scaled_default_thr', tile=23, H=64, W=64, scale=[(0.0001, 0.0)] * 6),
    dict(name='t80x72_scaled_mixed', tile=24, H=80, W=72, land=1, thr='reflectance', mode='ignore',
         scale=[(0.0001, 0.0), (0.0001, 10.0), (0.0002, -50.0), (0.0001, 3.5), (0.00005, 0.0), (0.0001, -0.25)]),
    # round 6: unusual fill values through the reference's own `image == fill_value` test (:2195-2209) --
    # fills at 0 (which the clip turns into 1 AFTER the test), at both ends of int16, at 1 (a legal reflectance), a
    # Fmask fill of 0 (the most common Fmask value); and NaN fills (no nodata value that any pixel can equal: the test is
    # disabled for that plane), mixed with ordinary ones, with a Fmask fill of NaN
    dict(name='t64_fills_zero_max', tile=25, H=64, W=64, land=1, ocean=1, fills='zero_max'),
    dict(name='t72x40_fills_nan', tile=26, H=72, W=40, shad=1, fills='nan'),
]


def _corridor_tile(H, W):
    """Bands and Fmask of the 'corridors' case: every pixel valid; land = the survey's known answer
    (500,600,700,3000,2500,1500) -> WTR-1 0, water = (300,400,300,200,100,50) -> DIAG 11111 / WTR-1 1."""
    land_px, water_px = (500, 600, 700, 3000, 2500, 1500), (300, 400, 300, 200, 100, 50)
    water = np.zeros((H, W), bool)
    fm = np.zeros((H, W), np.uint8)
    ADJ, SNOW, CLOUDBIT = 4, 16, 2

    def stroke(cells, kind, L):
        # cells: the corridor's pixels in order; kind selects seed / surface / obstacle
        cells = [(y, x) for y, x in cells if 0 <= y < H and 0 <= x < W]
        for y, x in cells:
            fm[y, x] |= ADJ
            if kind in (1, 2, 5):
                water[y, x] = True
        if not cells:
            return
        if kind in (0, 1, 3, 4):                      # snow seed ON the first corridor pixel
            fm[cells[0]] |= SNOW
        if kind in (2, 5):                            # seed in the middle
            fm[cells[len(cells) // 2]] |= SNOW
        if kind == 3 and len(cells) > 6:              # a cloud pixel in the way (CLOUD != 0: not in the area)
            fm[cells[6]] |= CLOUDBIT
        if kind == 4 and len(cells) > 3:              # water only on the far half: the second dilation's mask
            for y, x in cells[len(cells) // 2:]:
                water[y, x] = True
        if kind == 5:                                 # and a second seed at the far end
            fm[cells[-1]] |= SNOW
    lengths = [1, 2, 5, 7, 8, 9, 10, 11, 12, 15, 17, 18, 25]
    n = 0
    # horizontal corridors in the upper half, three rows apart (4-neighbourhood: they do not touch)
    for y in range(2, 128, 3):
        x = 3 + (n * 37) % 60
        while x < W + 10:
            L, kind = lengths[n % len(lengths)], n % 6
            stroke([(y, x + i) for i in range(L)], kind, L)
            x += L + 4 + (n % 3)
            n += 1
    # vertical corridors in the lower half, three columns apart; they cross y = 222 and run into the bottom edge
    for x in range(2, W, 3):
        y = 132 + (n * 29) % 40
        while y < H + 10:
            L, kind = lengths[n % len(lengths)], n % 6
            stroke([(y + i, x) for i in range(L)], kind, L)
            y += L + 4 + (n % 3)
            n += 1
    # a few two-pixel-wide and L-shaped corridors over the horizontal ones' right margin are not needed: widen some
    for y in range(2, 128, 12):
        fm[y + 1, 40:70] |= fm[y, 40:70] & ADJ       # two rows wide where the row above is a corridor
    bands = [np.where(water, w, l).astype(np.int16) for l, w in zip(land_px, water_px)]
    return bands, fm


def gen_tiles(ref):
    rng = np.random.default_rng(7)
    only = os.environ.get('GOLDEN_ONLY_TILES')           # e.g. 'cover': regenerate a subset, leave the rest untouched
    for case in TILE_CASES:
        if only and only not in case['name']:
            continue
        H, W = case['H'], case['W']
        s = synth_tile(case['tile'], H, W, with_masks=True)
        bands = [b.copy() for b in s['bands']]
        fmask = s['fmask'].copy()
        band_fills = [-9999.0] * 6
        fmask_fill = 255.0
        if case.get('fills') == 'other':
            # fill only present in SOME bands, different value per band
            band_fills = [-9999.0, -1000.0, 0.0, -9999.0, 32767.0, -9999.5]
            fmask_fill = 64.0
            for b, f in zip(bands, band_fills):
                if float(f).is_integer():
                    sel = rng.random(b.shape) < 0.02
                    b[sel] = int(f)
        if case.get('fills') in ('zero_max', 'nan'):
            frng = np.random.default_rng(2000 + case['tile'])       # (its own stream: the order of the cases above is frozen)
            if case['fills'] == 'zero_max':
                band_fills = [0.0, 32767.0, 1.0, -32768.0, 0.0, -1.0]
                fmask_fill = 0.0
            else:
                band_fills = [float('nan'), -9999.0, float('nan'), float('nan'), 1000.0, float('nan')]
                fmask_fill = float('nan')
            for b, f in zip(bands, band_fills):
                sel = frng.random(b.shape) < 0.03
                b[sel] = -9999 if f != f else int(f)                # planes with a NaN fill still hold -9999s: now plain data
        if case.get('mode') == 'cover':
            # make spatially coherent snow / adjacent blobs so the dilation matters
            yy, xx = np.mgrid[0:H, 0:W]
            blob = ((yy // 9 + xx // 11) % 5 == 0)
            fmask = np.where(blob & (fmask != 255), fmask | 4, fmask & ~np.uint8(4)).astype(np.uint8)
            snowb = ((yy // 7 + 2 * (xx // 5)) % 9 == 0)
            if case.get('blobs') == 'discs':
                # wide adjacent-to-cloud rings with cloud-free interiors and snow fields touching them
                crng = np.random.default_rng(1000 + case['tile'])
                blob = np.zeros((H, W), bool)
                snowb = np.zeros((H, W), bool)
                for _ in range(max(4, H * W // 9000)):
                    cy, cx, r = crng.integers(0, H), crng.integers(0, W), crng.integers(8, 40)
                    d2 = (yy - cy) ** 2 + (xx - cx) ** 2
                    blob |= (d2 < r * r) & (d2 >= (r - crng.integers(3, 16)) ** 2)
                    cy2, cx2 = cy + crng.integers(-r, r + 1), cx + crng.integers(-r, r + 1)
                    snowb |= (yy - cy2) ** 2 + (xx - cx2) ** 2 < crng.integers(2, 9) ** 2
                # adjacent pixels are cloud / shadow free in Fmask (so that CLOUD == 0 there)
                fmask = np.where(blob & (fmask != 255), fmask & ~np.uint8(2 | 8), fmask).astype(np.uint8)
                fmask = np.where(blob & (fmask != 255), fmask | 4, fmask & ~np.uint8(4)).astype(np.uint8)
            fmask = np.where(snowb & (fmask != 255), fmask | 16, fmask).astype(np.uint8)
            if case.get('blobs') == 'corridors':
                bands, fmask = _corridor_tile(H, W)
        land = s['land'] if case.get('land') else None
        shad = s['shad'].astype(bool) if case.get('shad') else None
        ocean = s['ocean'] if case.get('ocean') else None
        mode = case.get('mode', 'mask')
        thr_tag = case.get('thr', 'default')
        thr = make_thresholds(ref, **ALT_THRESHOLDS[thr_tag])
        lists = CUSTOM_AEROSOL if case.get('lists') == 'custom' else DEFAULT_AEROSOL
        res = run_reference_chain(ref, bands, fmask, thr, land, shad, ocean, mode,
                                  case.get('aerosol', True), lists,
                                  band_fills, fmask_fill, offset_and_scale=case.get('scale'))
        store = {'in_bands': np.stack(bands), 'in_fmask': fmask,
                 'thr': np.array([getattr(thr, k) for k in THR_KEYS], dtype=np.float64),
                 'band_fills': np.array(band_fills), 'fmask_fill': np.array(fmask_fill),
                 'mode': np.array(mode), 'apply_aerosol': np.array(case.get('aerosol', True)),
                 'aerosol_lists': np.array([','.join(map(str, l)) for l in lists])}
        if case.get('scale'):
            store['offset_and_scale'] = np.array(case['scale'], dtype=np.float64)      # [6][2]: scale_factor, add_offset
        if land is not None:
            store['in_land'] = land
        if shad is not None:
            store['in_shad'] = shad
        if ocean is not None:
            store['in_ocean'] = ocean
        for k, v in res.items():
            store['out_' + k] = v
        np.savez_compressed(os.path.join(GOLDEN, f"tile_{case['name']}.npz"), **store)
        print('tile', case['name'], H, W)


SHADOW_CASES = [
    # name, tile, H, W (with margin), sun azimuth, sun elevation, min slope, max inc
    dict(name='s_default', tile=0, H=220, W=260, az=143.2, el=55.5, mn=-5, mx=40),
    dict(name='s_low_sun', tile=1, H=180, W=200, az=231.7, el=18.25, mn=-5, mx=40),
    dict(name='s_noon_north', tile=2, H=150, W=170, az=0.0, el=89.0, mn=-5, mx=40),
    dict(name='s_other_thresholds', tile=3, H=160, W=160, az=95.0, el=33.0, mn=2.5, mx=55.5),
    dict(name='s_thin', tile=4, H=102, W=140, az=310.0, el=40.0, mn=-5, mx=40),
    # terraced DEMs (heights rounded to whole metres: few distinct slopes, many pixels each) with both
    # thresholds ON frequent float32 angle values, so that hundreds of pixels sit where float32 and
    # float64 arithmetic part: these are the cases on which the two promotion regimes give different layers
    dict(name='s_terraced_flat_tie', tile=5, H=200, W=240, az=143.2, el=55.5, terraced=True,
         mn=0.5720064043998718, mx=34.499996185302734),
    dict(name='s_terraced_low_sun', tile=6, H=180, W=180, az=231.7, el=18.25, terraced=True,
         mn=0.5918244123458862, mx=72.34337615966797),
    dict(name='s_terraced_high_sun', tile=8, H=160, W=200, az=10.0, el=70.0, terraced=True,
         mn=0.9403377175331116, mx=20.940963745117188),
]


def _shadow_dem(case):
    from proteus_amd.synth import synth_dem
    dem = synth_dem(case['tile'], case['H'], case['W'])
    return np.round(dem).astype(np.float32) if case.get('terraced') else dem


def gen_shadow(ref):
    from proteus_amd.synth import synth_dem
    margin = ref.DEM_MARGIN_IN_PIXELS
    for case in SHADOW_CASES:
        dem = _shadow_dem(case)
        full = ref._compute_opera_shadow_layer(dem, case['az'], case['el'], case['mn'], case['mx'])
        cropped = ref._crop_2d_array_all_sides(full, margin)
        np.savez_compressed(os.path.join(GOLDEN, f"shadow_{case['name']}.npz"), dem=dem,
                            az=np.array(case['az']), el=np.array(case['el']),
                            mn=np.array(case['mn']), mx=np.array(case['mx']),
                            margin=np.array(margin), full=full, cropped=cropped,
                            numpy_version=np.array(np.__version__))
        print('shadow', case['name'], full.dtype, float(full.mean()))


class _WeakScalarNumpy:
    """numpy as the reference module sees it while the 'legacy' shadow goldens are made: every
    attribute is numpy's own, except that radians / sin / cos / sqrt of a SCALAR return a Python float
    instead of an np.float64.

    Why that reproduces numpy 1.23.5 (which the reference pins, setup.py:78) under numpy >= 2: the only
    place the two promotion regimes part in _compute_opera_shadow_layer :4246-4281 is where a float32
    ARRAY meets one of the float64 SCALARS derived from the sun angles.  numpy < 2 (value-based casting)
    keeps such an operation in float32, with the scalar converted to float32 first; numpy >= 2 (NEP 50)
    makes it float64 -- unless the scalar is a Python float, which NEP 50 treats as weak: float32 loop,
    scalar converted to float32 first, i.e. exactly what value-based casting did.  Scalar-with-scalar
    arithmetic is double precision in all three cases.  So the reference's OWN code runs, expression by
    expression, with the arithmetic numpy 1.23.5 would give it; only the scalar type is swapped."""
    _SCALAR_FUNCS = ('radians', 'sin', 'cos')

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        attr = getattr(self._real, name)
        if name in self._SCALAR_FUNCS:
            def weak(x, *a, **k):
                out = attr(x, *a, **k)
                return float(out) if np.ndim(out) == 0 else out
            return weak
        return attr


def gen_shadow_legacy(ref):
    """'legacy' fixtures: the reference's _compute_opera_shadow_layer itself, executed with
    _WeakScalarNumpy in place of its module-level `np` (restored afterwards)."""
    from proteus_amd.synth import synth_dem
    margin = ref.DEM_MARGIN_IN_PIXELS
    real = ref.np
    ref.np = _WeakScalarNumpy(real)
    try:
        for case in SHADOW_CASES:
            dem = _shadow_dem(case)
            full = ref._compute_opera_shadow_layer(dem, case['az'], case['el'], case['mn'], case['mx'])
            assert full.dtype == np.bool_
            cropped = ref._crop_2d_array_all_sides(full, margin)
            np.savez_compressed(os.path.join(GOLDEN, f"shadow_legacy_{case['name']}.npz"), dem=dem,
                                az=np.array(case['az']), el=np.array(case['el']),
                                mn=np.array(case['mn']), mx=np.array(case['mx']),
                                margin=np.array(margin), full=full, cropped=cropped,
                                numpy_version=np.array(np.__version__),
                                promotion=np.array('numpy < 2 value-based casting, emulated by weak scalars'))
            print('shadow legacy', case['name'], float(full.mean()))
    finally:
        ref.np = real


def gen_browse(ref):
    codes = np.arange(256, dtype=np.uint8).reshape(1, -1)
    out = {'codes': codes}
    for mask in range(32):
        collapse, excl, nw, cl, sn = [(mask >> k) & 1 == 1 for k in range(5)]
        for ocean in (True, False):
            key = f'b_{int(collapse)}{int(excl)}{int(nw)}{int(cl)}{int(sn)}{int(ocean)}'
            out[key] = ref._compute_browse_array(
                codes, flag_collapse_wtr_classes=collapse, exclude_psw_aggressive=excl,
                set_not_water_to_nodata=nw, set_cloud_to_nodata=cl, set_snow_to_nodata=sn,
                set_ocean_masked_to_nodata=ocean)
    np.savez_compressed(os.path.join(GOLDEN, 'browse_tables.npz'), **out)
    print('browse_tables.npz', len(out) - 1, 'option sets')


LAND_CASES = [dict(name='l_standard', tile=0, H=90, W=120, year=2021, kind='standard'),
              dict(name='l_water_heavy', tile=1, H=64, W=64, year=2020, kind='water heavy'),
              dict(name='l_no_forest', tile=2, H=50, W=70, year=2000, kind='standard', forest=[]),
              dict(name='l_odd', tile=3, H=33, W=41, year=2099, kind='standard')]
DEFAULT_FOREST = [20, 50, 111, 113, 115, 116, 121, 123, 125, 126]


def gen_landcover(ref):
    """Replays create_landcover_mask :994-1115 on already-warped arrays with the
    reference's own helpers (the two gdal.Warp calls are the part that cannot run here)."""
    from proteus_amd.synth import synth_landcover_inputs
    cls = ref.dswx_hls_landcover_classes_dict
    for case in LAND_CASES:
        wc, cg = synth_landcover_inputs(case['tile'], case['H'], case['W'])
        forest_classes = case.get('forest', DEFAULT_FOREST)
        water = ref.decimate_by_summation(np.isin(wc, [80, 90, 95]).astype(np.uint8), 3, 3)
        urban = ref.decimate_by_summation((wc == 50).astype(np.uint8), 3, 3)
        tree = ref.decimate_by_summation((wc == 10).astype(np.uint8), 3, 3)
        forest = np.zeros_like(tree, dtype=np.uint8)
        for c in forest_classes:
            forest |= (cg == c)
        tree = np.where(forest, tree, 0)
        land = np.full(water.shape, cls['fill_value'], dtype=np.uint8)
        thr = ref.landcover_threshold_dict[case['kind']]
        off = case['year'] - 2000
        ref._update_landcover_array(land, tree, thr[0], cls['evergreen_forest'])
        ref._update_landcover_array(land, urban, thr[1], cls['low_intensity_developed_offset'] + off)
        ref._update_landcover_array(land, urban, thr[2], cls['high_intensity_developed_offset'] + off)
        ref._update_landcover_array(land, water, thr[3], cls['water'])
        np.savez_compressed(os.path.join(GOLDEN, f"land_{case['name']}.npz"), worldcover_up3=wc,
                            copernicus=cg, forest_classes=np.array(forest_classes, dtype=np.int32),
                            thresholds=np.array(thr, dtype=np.int32), year=np.array(case['year']),
                            kind=np.array(case['kind']), land=land)
        u, c = np.unique(land, return_counts=True)
        print('land', case['name'], dict(zip(u.tolist(), c.tolist())))


def main():
    ref = import_reference()
    if ref is None:
        raise SystemExit('reference tree not present; goldens can only be '
                         'regenerated in the build container')
    os.makedirs(GOLDEN, exist_ok=True)
    which = sys.argv[1:] or ['tables', 'diag', 'tiles', 'shadow', 'browse', 'land']
    if 'browse' in which:
        gen_browse(ref)
    if 'land' in which:
        gen_landcover(ref)
    if 'tables' in which:
        gen_tables(ref)
    if 'diag' in which:
        gen_diag(ref)
    if 'tiles' in which:
        gen_tiles(ref)
    if 'shadow' in which:
        gen_shadow(ref)
        gen_shadow_legacy(ref)


if __name__ == '__main__':
    main()
