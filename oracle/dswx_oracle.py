"""TEST INFRASTRUCTURE -- CPU oracle for the DSWx-HLS per-pixel path (numpy).

This file is a *checker*.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import it; the product (proteus_amd/) never
does and has no CPU fallback.

It restates, whole-array numpy op for whole-array numpy op (same dtypes, same
int16 wrap-around, same float64 quotients, same order of masked assignments), the
per-pixel chain of PROTEUS `src/proteus/dswx_hls.py`.  Each function cites the
reference lines it follows.  Parity pinning: `tests/test_oracle_golden.py` checks
every function here against tests/golden/*.npz, which `oracle/gen_golden.py`
produced by running the reference's own functions (imported from /root/reference
in the build container) -- including the reference's unit-test vector
(tests/test_dswx_hls_units.py:7-28) and the survey's known answers.

Because it keeps the reference's full-array temporaries it is also what bench.py
times as the "reference CPU/numpy path" (`cpu_baseline.kind == "port"`).
"""
import numpy as np

# ---------------------------------------------------------------------------
# constants (src/proteus/dswx_hls.py:26-31, 45-46, 49, 94-95, 146-185)
# ---------------------------------------------------------------------------
FILL_U8 = 255                      # UINT8_FILL_VALUE :49
DIAG_FILL_DECIMAL = 0b100000       # DIAGNOSTIC_LAYER_NO_DATA_DECIMAL :94
DIAG_FILL_BINARY_REPR = 65535      # DIAGNOSTIC_LAYER_NO_DATA_BINARY_REPR :95
AEROSOL_MAX_NIR = 0.1 / 0.0001     # AEROSOL_REMAPPING_MAX_NIR :45-46
WTR_SNOW, WTR_CLOUD, WTR_OCEAN = 252, 253, 254       # :160-162
LAND_WATER, LAND_EVERGREEN = 200, 201                # :252-264
LAND_LOW_DEV0, LAND_HIGH_DEV0 = 0, 100

# DIAG (5 test bits) -> WTR-1 class, interpreted_dswx_band_dict :97-143
_NOT_WATER = (0b00000, 0b00001, 0b00010, 0b00100, 0b01000)
_HIGH_CONF = (0b01111, 0b10111, 0b11011, 0b11101, 0b11110, 0b11111)
_MODERATE = (0b00111, 0b01011, 0b01101, 0b01110, 0b10011, 0b10101, 0b10110,
             0b11001, 0b11010, 0b11100)
_PSW_CONSERVATIVE = (0b11000,)
_PSW_AGGRESSIVE = (0b00011, 0b00101, 0b00110, 0b01001, 0b01010, 0b01100,
                   0b10000, 0b10001, 0b10010, 0b10100)
DIAG_TO_CLASS = {}
for _cls, _keys in enumerate((_NOT_WATER, _HIGH_CONF, _MODERATE,
                              _PSW_CONSERVATIVE, _PSW_AGGRESSIVE)):
    for _k in _keys:
        DIAG_TO_CLASS[_k] = _cls
DIAG_TO_CLASS[DIAG_FILL_DECIMAL] = FILL_U8

# collapse_wtr_classes_dict :201-213
COLLAPSE = {0: 0, 1: 1, 2: 1, 3: 2, 4: 2,
            WTR_OCEAN: WTR_OCEAN, WTR_SNOW: WTR_SNOW, WTR_CLOUD: WTR_CLOUD,
            FILL_U8: FILL_U8}

# defaults/dswx_hls.yaml:176-212
DEFAULT_THRESHOLDS = dict(
    wigt=0.124, awgt=0.0,
    pswt_1_mndwi=-0.44, pswt_1_nir=1500, pswt_1_swir1=900, pswt_1_ndvi=0.7,
    pswt_2_mndwi=-0.5, pswt_2_blue=1000, pswt_2_nir=2500, pswt_2_swir1=3000,
    pswt_2_swir2=1000, lcmask_nir=1200)

# defaults/dswx_hls.yaml:77-89; key = WTR-1 class being remapped to class 1
DEFAULT_AEROSOL_FMASK_VALUES = {
    0: [224, 160, 96],
    2: [224, 160, 96],
    3: [224, 192, 160, 128, 96],
    4: [224, 192, 160, 128, 96]}


class Thresholds:
    """Same twelve attributes as HlsThresholds (src/proteus/dswx_hls.py:274-318)."""

    def __init__(self, **kw):
        vals = dict(DEFAULT_THRESHOLDS)
        vals.update(kw)
        for k, v in vals.items():
            if k not in DEFAULT_THRESHOLDS:
                raise KeyError(k)
            setattr(self, k, v)


# ---------------------------------------------------------------------------
# A0  input conditioning  (_load_hls_band_from_file :2195-2209, :2224-2226, :2298-2299)
# ---------------------------------------------------------------------------
def condition_inputs(bands, fmask, band_fills=(-9999.,) * 6, fmask_fill=255.,
                     clip_negative_reflectance=True, offset_and_scale=None):
    """bands: six int16 arrays (blue, green, red, nir, swir1, swir2).

    Returns (clipped_bands, invalid) with `invalid` the cumulative
    `image == fill_value` over the six bands and Fmask; reflectances (not
    Fmask) are then clipped to >= 1.  offset_and_scale: six (scale_factor,
    add_offset) pairs = flag_offset_and_scale_inputs (:2300-2302): each clipped
    band becomes `scale_factor * (float32(image) - offset)` and the whole chain
    then runs on float32 arrays.
    """
    invalid = None
    for img, fill in list(zip(bands, band_fills)) + [(fmask, fmask_fill)]:
        if fill is None:
            continue
        eq = img == fill
        invalid = eq if invalid is None else np.logical_or(invalid, eq)
    if invalid is None:
        invalid = np.zeros(fmask.shape, dtype=bool)
    if clip_negative_reflectance:
        bands = [np.clip(img, 1, None) for img in bands]
    if offset_and_scale is not None:
        bands = [float(sf) * (np.asarray(img, dtype=np.float32) - float(off))       # :2300-2302, Python-float scalars
                 for img, (sf, off) in zip(bands, offset_and_scale)]
    return list(bands), invalid


# ---------------------------------------------------------------------------
# A2  _compute_preliminary_cloud_layer :1919-1993
# ---------------------------------------------------------------------------
def compute_preliminary_cloud_layer(fmask, mask_adjacent_to_cloud_mode):
    if mask_adjacent_to_cloud_mode not in ('mask', 'ignore', 'cover'):
        raise Exception('ERROR mask adjacent to cloud/cloud-shadow mode:'
                        f' {mask_adjacent_to_cloud_mode}')
    cloud = np.zeros(fmask.shape, dtype=np.uint8)
    cloud[np.bitwise_and(fmask, 8) == 8] = 1          # Fmask bit 3: cloud shadow
    if mask_adjacent_to_cloud_mode == 'mask':
        cloud[np.bitwise_and(fmask, 4) == 4] = 1      # bit 2: adjacent
    cloud[np.bitwise_and(fmask, 2) == 2] += 4         # bit 1: cloud
    return cloud


# ---------------------------------------------------------------------------
# A3  coverage counters (generate_dswx_layers :5104-5136)
# ---------------------------------------------------------------------------
def coverage_counters(invalid, preliminary_cloud, ocean_mask=None):
    """Returns dict with the three counts and the three floor-percent fields."""
    total = invalid.size
    valid = ~invalid
    if ocean_mask is not None:
        valid = np.logical_and(valid, ocean_mask)
        n_not_ocean = int(np.sum(ocean_mask))
    else:
        n_not_ocean = total
    n_valid = int(np.sum(valid))
    n_cloud_and_valid = int(np.sum((preliminary_cloud != 0) & valid))
    spatial = int(100 * float(n_valid) / total) if total else 0
    cloud_cov = 0 if n_valid == 0 else int(100 * float(n_cloud_and_valid) / n_valid)
    spatial_no_ocean = 0 if n_not_ocean == 0 else \
        int(100 * float(n_valid) / n_not_ocean)
    return dict(n_valid=n_valid, n_cloud_and_valid=n_cloud_and_valid,
                n_not_ocean=n_not_ocean, SPATIAL_COVERAGE=spatial,
                CLOUD_COVERAGE=cloud_cov,
                SPATIAL_COVERAGE_EXCLUDING_MASKED_OCEAN=spatial_no_ocean)


# ---------------------------------------------------------------------------
# A4  _compute_diagnostic_tests :1840-1916
# ---------------------------------------------------------------------------
def spectral_indices(blue, green, red, nir, swir1, swir2):
    """MNDWI, MBSRV, MBSRN, AWESH, NDVI exactly as :1872-1887 forms them:
    int16 sums/differences wrap, quotients and AWESH are float64."""
    with np.errstate(divide='ignore', invalid='ignore'):
        mndwi = (green - swir1) / (green + swir1)
        mbsrv = green + red
        mbsrn = nir + swir1
        awesh = blue + (2.5 * green) - (1.5 * mbsrn) - (0.25 * swir2)
        ndvi = (nir - red) / (nir + red)
    return mndwi, mbsrv, mbsrn, awesh, ndvi


def compute_diagnostic_tests(blue, green, red, nir, swir1, swir2, thr):
    mndwi, mbsrv, mbsrn, awesh, ndvi = spectral_indices(
        blue, green, red, nir, swir1, swir2)
    diag = np.zeros(blue.shape, dtype=np.uint16)
    with np.errstate(invalid='ignore'):
        diag[mndwi > thr.wigt] += 1                               # test 1
        diag[mbsrv > mbsrn] += 2                                  # test 2
        diag[awesh > thr.awgt] += 4                               # test 3
        sel = np.where((mndwi > thr.pswt_1_mndwi) &               # test 4
                       (swir1 < thr.pswt_1_swir1) &
                       (nir < thr.pswt_1_nir) &
                       (ndvi < thr.pswt_1_ndvi))
        diag[sel] += 8
        sel = np.where((mndwi > thr.pswt_2_mndwi) &               # test 5
                       (blue < thr.pswt_2_blue) &
                       (swir1 < thr.pswt_2_swir1) &
                       (swir2 < thr.pswt_2_swir2) &
                       (nir < thr.pswt_2_nir))
        diag[sel] += 16
    return diag


# ---------------------------------------------------------------------------
# A6  generate_interpreted_layer :1687-1707
# ---------------------------------------------------------------------------
def generate_interpreted_layer(diag_decimal):
    out = np.full(diag_decimal.shape, FILL_U8, dtype=np.uint8)
    for key, value in DIAG_TO_CLASS.items():
        out[diag_decimal == key] = value
    return out


# ---------------------------------------------------------------------------
# A7  _get_binary_representation :4286-4317
# ---------------------------------------------------------------------------
def get_binary_representation(diag_decimal, nbits=6):
    out = np.zeros_like(diag_decimal, dtype=np.uint16)
    rest = diag_decimal
    for i in range(nbits):
        rest, bit = np.divmod(rest, 2)
        if i < 5:
            out += bit * (10 ** i)
        else:
            out[np.where(bit)] = DIAG_FILL_BINARY_REPR
    return out


# ---------------------------------------------------------------------------
# A9  _apply_aerosol_class_remapping :1249-1302 (+ _single_class :1210-1246)
# ---------------------------------------------------------------------------
def apply_aerosol_class_remapping(wtr_1, nir, preliminary_cloud, fmask,
                                  fmask_values_by_class=None):
    """In place on wtr_1 and preliminary_cloud, classes visited 0, 2, 3, 4."""
    if fmask_values_by_class is None:
        fmask_values_by_class = DEFAULT_AEROSOL_FMASK_VALUES
    for in_class in (0, 2, 3, 4):
        remap = (np.isin(fmask, fmask_values_by_class[in_class]) &
                 (wtr_1 == in_class) &
                 (nir <= AEROSOL_MAX_NIR))
        wtr_1[remap] = 1
        sel = np.where(remap & (preliminary_cloud != FILL_U8))
        preliminary_cloud[sel] = np.bitwise_or(preliminary_cloud[sel], 8)


# ---------------------------------------------------------------------------
# A10  _apply_landcover_and_shadow_masks :1305-1378, predicates :1133-1207
# ---------------------------------------------------------------------------
def apply_landcover_and_shadow_masks(wtr_1, nir, landcover, shadow, thr):
    out = wtr_1.copy()
    is_water_class = (wtr_1 >= 1) & (wtr_1 <= 4)
    if shadow is not None and landcover is None:
        out[np.where((shadow == 0) & is_water_class)] = 0
    elif shadow is not None:
        out[np.where((shadow == 0) & (~(landcover == LAND_WATER)) &
                     is_water_class)] = 0
    if landcover is None:
        return out
    is_psw = (wtr_1 == 3) | (wtr_1 == 4)
    bright_nir = nir > thr.lcmask_nir
    out[np.where((landcover == LAND_EVERGREEN) & bright_nir & is_psw)] = 0
    low_dev = (landcover >= LAND_LOW_DEV0) & (landcover < LAND_LOW_DEV0 + 100)
    out[np.where(low_dev & bright_nir & is_psw)] = 0
    high_dev = (landcover >= LAND_HIGH_DEV0) & (landcover < LAND_HIGH_DEV0 + 100)
    out[np.where(high_dev & is_water_class)] = 0
    return out


# ---------------------------------------------------------------------------
# A11  _add_snow_to_cloud_layer :1996-2086  ('cover' branch :2055-2078)
# ---------------------------------------------------------------------------
def add_snow_to_cloud_layer(wtr_2, cloud, fmask, mask_adjacent_to_cloud_mode):
    """In place on `cloud`; returns it."""
    snow = np.bitwise_and(fmask, 16) == 16
    if mask_adjacent_to_cloud_mode == 'cover':
        from scipy.ndimage import binary_dilation
        adjacent = np.bitwise_and(fmask, 4) == 4
        grow_area = adjacent & (cloud == 0)
        snow = binary_dilation(snow, iterations=10, mask=grow_area)
        grow_area &= ((wtr_2 >= 1) & (wtr_2 <= 4))
        clear = (~snow) & (cloud == 0)
        clear = binary_dilation(clear, iterations=7, mask=grow_area)
        snow[clear] = False
    cloud[snow] += 2
    cloud[wtr_2 == FILL_U8] = FILL_U8
    return cloud


# ---------------------------------------------------------------------------
# A12  _apply_cloud_masking :2089-2133
# ---------------------------------------------------------------------------
def apply_cloud_masking(wtr_2, cloud):
    wtr = wtr_2.copy()
    wtr[(cloud != 0) & (cloud != 8)] = WTR_CLOUD
    wtr[(cloud == 2) | (cloud == 10)] = WTR_SNOW
    wtr[wtr_2 == WTR_OCEAN] = WTR_OCEAN
    wtr[wtr_2 == FILL_U8] = FILL_U8
    return wtr


# ---------------------------------------------------------------------------
# A13  _get_binary_water_layer :1710-1730
# ---------------------------------------------------------------------------
def get_binary_water_layer(wtr):
    bwtr = wtr.copy()
    for c in range(1, 5):
        bwtr[wtr == c] = 1
    return bwtr


# ---------------------------------------------------------------------------
# A14  _get_confidence_layer :1733-1837
# ---------------------------------------------------------------------------
def get_confidence_layer(wtr_2, cloud):
    conf = wtr_2.copy()
    cloudy = np.isin(cloud, [1, 3, 4, 5, 6, 7, 9, 11, 12, 13, 14, 15])
    for c in range(5):
        conf[(conf == c) & cloudy] = 10 + c
    snowy = cloud == 2
    for c in range(5):
        conf[(conf == c) & snowy] = 20 + c
    return conf


# ---------------------------------------------------------------------------
# A15  _collapse_wtr_classes :2578-2598
# ---------------------------------------------------------------------------
def collapse_wtr_classes(layer):
    out = np.full_like(layer, FILL_U8)
    for src, dst in COLLAPSE.items():
        out[layer == src] = dst
    return out


# ---------------------------------------------------------------------------
# A16  the hot segment of generate_dswx_layers :5088-5112, :5225-5286, :5358-5369
# ---------------------------------------------------------------------------
def classify_tile(bands, fmask, thr=None, *, landcover=None, shadow=None,
                  ocean_mask=None, band_fills=(-9999.,) * 6, fmask_fill=255.,
                  clip_negative_reflectance=True,
                  mask_adjacent_to_cloud_mode='mask',
                  apply_aerosol=True, aerosol_fmask_values=None,
                  collapse=True, with_indices=False, offset_and_scale=None):
    """Run the whole per-pixel chain on one tile, in the reference's order.

    `bands` are the RAW int16 planes as read from file (fill values still in
    place).  Returns a dict of the layers in the form the reference SAVES them
    when `collapse` is True (WTR, WTR-1, WTR-2 collapsed at save time,
    :2688-2689), or in the in-memory uncollapsed form otherwise, plus
    `WTR-1-AEROSOL` (the in-place remapped WTR-1 that feeds WTR-2, and that the
    multi-band output file receives, :5381-5396) and `counters`.
    """
    if thr is None:
        thr = Thresholds()
    (blue, green, red, nir, swir1, swir2), invalid = condition_inputs(
        bands, fmask, band_fills, fmask_fill, clip_negative_reflectance, offset_and_scale)
    invalid_ind = np.where(invalid)

    cloud = compute_preliminary_cloud_layer(fmask, mask_adjacent_to_cloud_mode)
    counters = coverage_counters(invalid, cloud, ocean_mask)

    diag_decimal = compute_diagnostic_tests(blue, green, red, nir, swir1,
                                            swir2, thr)
    diag_decimal[invalid_ind] = DIAG_FILL_DECIMAL                  # :5227
    wtr_1 = generate_interpreted_layer(diag_decimal)               # :5229
    diag = get_binary_representation(diag_decimal)                 # :5231
    if ocean_mask is not None:
        wtr_1[ocean_mask == 0] = WTR_OCEAN                         # :5245
    wtr_1[invalid_ind] = FILL_U8                                   # :5249
    wtr_1_saved = wtr_1.copy()                                     # saved at :5251
    if apply_aerosol:
        apply_aerosol_class_remapping(wtr_1, nir, cloud, fmask,
                                      aerosol_fmask_values)        # :5261
    wtr_2 = apply_landcover_and_shadow_masks(wtr_1, nir, landcover, shadow,
                                             thr)                  # :5268
    cloud = add_snow_to_cloud_layer(wtr_2, cloud, fmask,
                                    mask_adjacent_to_cloud_mode)   # :5282
    wtr = apply_cloud_masking(wtr_2, cloud)                        # :5286
    bwtr = get_binary_water_layer(wtr)                             # :5358
    conf = get_confidence_layer(wtr_2, cloud)                      # :5368
    out = {'DIAG': diag, 'WTR-1': wtr_1_saved, 'WTR-1-AEROSOL': wtr_1,
           'WTR-2': wtr_2, 'WTR': wtr, 'BWTR': bwtr, 'CONF': conf,
           'CLOUD': cloud, 'counters': counters}
    if collapse:
        for name in ('WTR', 'WTR-1', 'WTR-1-AEROSOL', 'WTR-2'):
            out[name] = collapse_wtr_classes(out[name])
    if with_indices:
        mndwi, _, _, awesh, ndvi = spectral_indices(blue, green, red, nir,
                                                    swir1, swir2)
        out.update(MNDWI=mndwi, NDVI=ndvi, AWESH=awesh)
    return out


# ---------------------------------------------------------------------------
# f1  _compute_opera_shadow_layer :4215-4283  and  _crop_2d_array_all_sides :4320
# ---------------------------------------------------------------------------
def compute_opera_shadow_layer(dem, sun_azimuth_angle, sun_elevation_angle,
                               min_slope_angle, max_sun_local_inc_angle,
                               pixel_spacing_x=30, pixel_spacing_y=30, legacy_promotion=False):
    """Same expressions, hence the same float32/float64 promotion as the reference gets
    from whichever numpy runs it (numpy >= 2 here: the products with the float64 sun
    scalars are float64; numpy 1.23.5, which the reference pins, keeps them float32).
    legacy_promotion=True restates the numpy < 2 behaviour under numpy >= 2 by rounding the
    five sun scalars to float32 first, which is all that value-based casting does to these
    expressions.  No numpy < 2 is in this image; the restatement is pinned to fixtures made by the
    reference's OWN function run with weak (Python float) sun scalars, which is how numpy >= 2 is made to
    use the float32 loops of numpy 1.23.5 (oracle/gen_golden.py::gen_shadow_legacy, tests/golden/shadow_legacy_*)."""
    sun_azimuth = np.radians(sun_azimuth_angle)
    sun_zenith = np.radians(90 - sun_elevation_angle)
    to_sun = [np.sin(sun_azimuth) * np.sin(sun_zenith),
              np.cos(sun_azimuth) * np.sin(sun_zenith),
              np.cos(sun_zenith)]
    if legacy_promotion:
        to_sun = [np.float32(v) for v in to_sun]
        sun_azimuth = np.float64(sun_azimuth)
        sin_az, cos_az = np.float32(np.sin(sun_azimuth)), np.float32(np.cos(sun_azimuth))
    else:
        sin_az, cos_az = np.sin(sun_azimuth), np.cos(sun_azimuth)
    grad_y, grad_x = np.gradient(dem)
    normal = [-grad_x / pixel_spacing_x, -grad_y / - abs(pixel_spacing_y), 1]
    norm = np.sqrt(normal[0] ** 2 + normal[1] ** 2 + 1)
    sun_inc_angle_degrees = np.degrees(np.arccos(
        (normal[0] * to_sun[0] + normal[1] * to_sun[1] + normal[2] * to_sun[2]) / norm))
    directional_slope_angle = np.degrees(np.arctan(normal[0] * sin_az + normal[1] * cos_az))
    backslope = directional_slope_angle <= min_slope_angle
    low_inc = sun_inc_angle_degrees <= max_sun_local_inc_angle
    return low_inc | (~backslope)


def crop_2d_array_all_sides(arr, margin):
    return arr[margin:-margin, margin:-margin]


# ---------------------------------------------------------------------------
# f4  _compute_browse_array :3057-3129 (input: the UNCOLLAPSED WTR layer)
# ---------------------------------------------------------------------------
def compute_browse_array(wtr_uncollapsed, flag_collapse_wtr_classes=True,
                         exclude_psw_aggressive=False, set_not_water_to_nodata=False,
                         set_cloud_to_nodata=False, set_snow_to_nodata=False,
                         set_ocean_masked_to_nodata=True):
    out = wtr_uncollapsed.copy()
    if exclude_psw_aggressive:
        out[out == 4] = 0
    if flag_collapse_wtr_classes:
        out = collapse_wtr_classes(out)
    if set_not_water_to_nodata:
        out[out == 0] = FILL_U8
    if set_cloud_to_nodata:
        out[out == WTR_CLOUD] = FILL_U8
    if set_snow_to_nodata:
        out[out == WTR_SNOW] = FILL_U8
    if set_ocean_masked_to_nodata:
        out[out == WTR_OCEAN] = FILL_U8
    return out


# ---------------------------------------------------------------------------
# f3  create_landcover_mask :994-1115 after the GDAL warps; decimate_by_summation :874-904
# ---------------------------------------------------------------------------
LANDCOVER_THRESHOLDS = {'standard': [6, 3, 7, 3], 'water heavy': [6, 3, 7, 1]}   # :270-271


def decimate_by_summation(image, size_y, size_x):
    out = None
    for i in range(size_y):
        for j in range(size_x):
            part = image[i::size_y, j::size_x]
            if out is None:
                cur = np.copy(part)
                out = np.zeros_like(cur)
            else:
                cur[0:part.shape[0], 0:part.shape[1]] = part
            out += cur
    return out


def landcover_mask_from_warped(worldcover_up3, copernicus, forest_classes, mask_type='standard',
                               year=2000):
    water = decimate_by_summation(np.isin(worldcover_up3, [80, 90, 95]).astype(np.uint8), 3, 3)
    urban = decimate_by_summation((worldcover_up3 == 50).astype(np.uint8), 3, 3)
    tree = decimate_by_summation((worldcover_up3 == 10).astype(np.uint8), 3, 3)
    forest = np.zeros_like(tree, dtype=np.uint8)
    if forest_classes is not None:
        for c in forest_classes:
            forest |= (copernicus == c)
    tree = np.where(forest, tree, 0)
    land = np.full(water.shape, FILL_U8, dtype=np.uint8)
    thr = LANDCOVER_THRESHOLDS[mask_type.lower()]
    off = year - 2000
    land[tree >= thr[0]] = LAND_EVERGREEN
    land[urban >= thr[1]] = LAND_LOW_DEV0 + off
    land[urban >= thr[2]] = LAND_HIGH_DEV0 + off
    land[water >= thr[3]] = LAND_WATER
    return land
