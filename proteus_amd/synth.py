"""Deterministic synthetic HLS tiles: host (numpy) generator.

Counter-based: every plane value is a pure function of (seed, tile index,
pixel index), so any tile of any batch can be regenerated anywhere.  The device
generator `dswx_synth_fill` (proteus_amd/csrc/dswx_hip.hip) implements the very
same integer recipe and is tested bit-for-bit against this file.

Recipe (all arithmetic modulo 2**64 / 2**32, no floating point):
  key = seed*K0 + tile*K1 + pixel ; h0 = mix(key) ; h1 = mix(h0+K0) ; h2 = mix(h1+K0)
  surface type from h0[0:16]; six reflectance noises from h1 (10 bits each);
  "value < 1" (clip) and "value > 16384" (int16 wrap) events from h0;
  Fmask bits, LAND class and SHAD from h2; OCEAN in 32-row bands from a
  per-band hash.  Fill pixels carry -9999 in every band and 255 in Fmask.
"""
import numpy as np

SEED = 20251010
K0 = 0x9E3779B97F4A7C15
K1 = 0xD1B54A32D192ED03
M0 = 0xBF58476D1CE4E5B9
M1 = 0x94D049BB133111EB
BAND_FILL = -9999
FMASK_FILL = 255

# type cut points on a 16-bit draw: water, wet, vegetation, bare, bright, fill
TYPE_CUTS = (14418, 26214, 42598, 55705, 64225)
# per type: six band means (blue, green, red, nir, swir1, swir2) and amplitude
TYPE_MEAN = ((350, 450, 350, 250, 150, 100),
             (500, 700, 600, 1300, 800, 500),
             (300, 600, 400, 3500, 1800, 900),
             (900, 1200, 1500, 2200, 2800, 2300),
             (6000, 6200, 6400, 6600, 3000, 2500))
TYPE_AMP = (300, 600, 600, 800, 2500)
LAND_CLASSES = (200, 201, 21, 121, 50, 150, 99, 100)

_U64 = np.uint64


def _mix(x):
    x = x ^ (x >> _U64(30))
    x = x * _U64(M0)
    x = x ^ (x >> _U64(27))
    x = x * _U64(M1)
    return x ^ (x >> _U64(31))


def _field(h, shift, bits):
    return ((h >> _U64(shift)) & _U64((1 << bits) - 1)).astype(np.int64)


def synth_tile(tile, height, width, seed=SEED, with_masks=False):
    """Returns dict: 'bands' (6 int16 [H,W], raw, fills in place), 'fmask' u8,
    and with `with_masks` also 'land', 'shad', 'ocean' (u8)."""
    n = height * width
    pix = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        key = _U64(seed) * _U64(K0) + _U64(tile) * _U64(K1) + pix
        h0 = _mix(key)
        h1 = _mix(h0 + _U64(K0))
        h2 = _mix(h1 + _U64(K0))

    draw = _field(h0, 0, 16)
    stype = np.searchsorted(np.asarray(TYPE_CUTS), draw, side='right')  # 0..5
    is_fill = stype == 5
    st = np.minimum(stype, 4)
    mean = np.asarray(TYPE_MEAN, dtype=np.int64)[st]          # [n, 6]
    amp = np.asarray(TYPE_AMP, dtype=np.int64)[st]            # [n]

    clip_evt = _field(h0, 16, 7) == 0
    clip_band = (_field(h0, 23, 3) * 6) >> 3
    clip_val = -_field(h0, 26, 8)
    wrap_evt = _field(h0, 34, 10) == 0

    bands = []
    for b in range(6):
        noise = _field(h1, 10 * b, 10)
        v = mean[:, b] + ((noise * 2 * amp) >> 10) - amp
        v = np.where(wrap_evt & ((b == 1) | (b == 4)), v + 19000, v)
        v = np.where(clip_evt & (clip_band == b), clip_val, v)
        v = np.where(is_fill, BAND_FILL, v)
        bands.append(v.astype(np.int16).reshape(height, width))

    aerosol = _field(h2, 0, 2)
    water = (_field(h2, 2, 5) < 10).astype(np.int64)      # p = .31
    snow = (_field(h2, 7, 5) < 2).astype(np.int64)        # p = .06
    shadow = (_field(h2, 12, 5) < 3).astype(np.int64)     # p = .09
    adjacent = (_field(h2, 17, 5) < 3).astype(np.int64)
    cloud = (_field(h2, 22, 5) < 4).astype(np.int64)      # p = .125
    cirrus = (_field(h2, 27, 5) < 1).astype(np.int64)
    fmask = (aerosol << 6) | (water << 5) | (snow << 4) | (shadow << 3) | \
        (adjacent << 2) | (cloud << 1) | cirrus
    fmask = np.where(is_fill, FMASK_FILL, fmask).astype(np.uint8)
    out = {'bands': bands, 'fmask': fmask.reshape(height, width)}
    if not with_masks:
        return out

    land_draw = _field(h2, 32, 8)
    land_cls = np.asarray(LAND_CLASSES, dtype=np.int64)[_field(h2, 40, 3)]
    land = np.where(land_draw < 179, 255, land_cls).astype(np.uint8)
    shad = (_field(h2, 43, 5) >= 3).astype(np.uint8)       # 0 (shadow) with p≈.09
    row_band = (pix // _U64(width)) >> _U64(5)
    with np.errstate(over='ignore'):
        hb = _mix(_U64(seed) * _U64(K1) + _U64(tile) * _U64(K0) + row_band +
                  _U64(0x5851F42D4C957F2D))
    ocean = (_field(hb, 0, 8) >= 13).astype(np.uint8)      # 0 (ocean) with p≈.05
    out.update(land=land.reshape(height, width),
               shad=shad.reshape(height, width),
               ocean=ocean.reshape(height, width))
    return out


def synth_dem(tile, height, width, seed=SEED):
    """Deterministic float32 DEM [height, width] in metres: a few sinusoidal ridges
    (slopes up to ~35 degrees at 30 m spacing), a flat lake and +-1.5 m of hash noise,
    so that both shadow tests and their thresholds are exercised."""
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    rng_phase = [((seed * 7919 + tile * 104729 + k * 1299709) % 6283) / 1000.0 for k in range(4)]
    z = (420.0 * np.sin(xx / 37.0 + rng_phase[0]) * np.cos(yy / 53.0 + rng_phase[1]) +
         180.0 * np.sin((xx + 2.0 * yy) / 19.0 + rng_phase[2]) +
         60.0 * np.cos((3.0 * xx - yy) / 11.0 + rng_phase[3]) + 900.0)
    z = np.where(z < 650.0, 650.0, z)                       # a flat "lake"
    pix = (yy * width + xx).astype(np.uint64)
    with np.errstate(over='ignore'):
        h = _mix(_U64(seed) * _U64(K1) + _U64(tile) * _U64(K0) + pix)
    noise = (_field(h, 0, 12).astype(np.float64) / 4096.0 - 0.5) * 3.0
    return (z + noise).astype(np.float32)


def synth_landcover_inputs(tile, height, width, seed=SEED):
    """(worldcover_up3 [3H,3W] u8, copernicus [H,W] u8): patchy class maps so that the
    3x3 counts cross every threshold of the LAND hierarchy."""
    wc_classes = np.array([10, 10, 10, 20, 30, 40, 50, 50, 60, 80, 90, 95, 100, 10, 50, 80], np.uint8)
    cg_classes = np.array([111, 113, 115, 116, 121, 123, 125, 126, 20, 50, 30, 40, 60, 80, 112, 200],
                          np.uint8)

    def hashed(h, w, cell, salt):
        yy, xx = np.mgrid[0:h, 0:w]
        key = ((yy // cell).astype(np.uint64) * _U64(40503) + (xx // cell).astype(np.uint64))
        with np.errstate(over='ignore'):
            hv = _mix(_U64(seed) * _U64(K0) + _U64(tile) * _U64(K1) + key + _U64(salt))
            fine = _mix(hv + (yy.astype(np.uint64) * _U64(w) + xx.astype(np.uint64)) * _U64(K0))
        return hv, fine

    hv, fine = hashed(3 * height, 3 * width, 7, 11)
    coarse_cls = wc_classes[_field(hv, 0, 4)]
    noisy_cls = wc_classes[_field(fine, 8, 4)]
    wc = np.where(_field(fine, 0, 3) < 3, noisy_cls, coarse_cls).astype(np.uint8)
    hv, fine = hashed(height, width, 9, 23)
    cg = np.where(_field(fine, 0, 3) < 1, cg_classes[_field(fine, 8, 4)],
                  cg_classes[_field(hv, 0, 4)]).astype(np.uint8)
    return wc, cg
