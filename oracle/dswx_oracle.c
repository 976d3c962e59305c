/*
 * TEST INFRASTRUCTURE -- scalar C restatement of the DSWx-HLS per-pixel chain.
 *
 * A checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load oracle/_build/libdswx_oracle.so.  It is
 * the fast comparator for full-size (3660x3660) tiles, where the numpy oracle
 * (oracle/dswx_oracle.py, pinned to the reference-generated goldens) takes ~8 s a
 * tile.  tests/test_oracle_golden.py pins THIS file to the same goldens.
 *
 * It follows PROTEUS src/proteus/dswx_hls.py literally, one pixel at a time, with
 * TRUE float64 division for MNDWI/NDVI (:1872, :1887) -- deliberately not the
 * division-free predicate the HIP kernel uses, so the two are independent.
 * Function comments give the reference lines.
 *
 * Also here: oracle_check_quotient_predicate(), an exhaustive (all 2^32 int16
 * pairs) proof-by-enumeration that the kernel's FMA predicate equals the
 * reference's `fl64(n/d) > t` / `< t`.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dswx_hip.h" /* dswx_params_t, plane structs: layout only */

static int16_t wrap16(int v) { return (int16_t)(uint16_t)(v & 0xffff); }

/* interpreted_dswx_band_dict :97-143 */
static const uint8_t k_interp[32] = {0, 0, 0, 4, 0, 4, 4, 2, 0, 4, 4, 2, 4, 2, 2, 1,
                                     4, 4, 4, 2, 4, 2, 2, 1, 3, 2, 2, 1, 2, 1, 1, 1};

/* _collapse_wtr_classes :2578-2598 */
static uint8_t collapse(uint8_t v) {
    switch (v) {
        case 0: return 0;
        case 1: case 2: return 1;
        case 3: case 4: return 2;
        case 252: case 253: case 254: case 255: return v;
        default: return 255;
    }
}

/*
 * The whole chain for `n` pixels of flat planes (any output may be NULL).
 * counters[3] += n_valid, n_cloud_and_valid, sum(ocean) (:5104-5112).
 * Returns 0, or -1 for an unsupported mode ('cover' is a neighbourhood op).
 */
int oracle_classify(const dswx_params_t* p, int64_t n, const dswx_planes_in_t* in,
                    const dswx_planes_out_t* out, int64_t* counters) {
    if (p->mask_adjacent_to_cloud_mode != DSWX_ADJ_MASK &&
        p->mask_adjacent_to_cloud_mode != DSWX_ADJ_IGNORE)
        return -1;
    int64_t n_valid = 0, n_cloud = 0, n_not_ocean = 0;
    for (int64_t i = 0; i < n; ++i) {
        int16_t v[6];
        int invalid = 0;
        /* _load_hls_band_from_file :2195-2209 (fill), :2298-2299 (clip) */
        for (int k = 0; k < 6; ++k) {
            v[k] = in->band[k][i];
            if ((double)v[k] == p->band_fill[k]) invalid = 1;
        }
        const uint8_t fm = in->fmask[i];
        if ((double)fm == p->fmask_fill) invalid = 1;
        if (p->clip_negative_reflectance)
            for (int k = 0; k < 6; ++k)
                if (v[k] < 1) v[k] = 1;
        const int16_t blue = v[0], green = v[1], red = v[2], nir = v[3], swir1 = v[4], swir2 = v[5];
        /* _compute_diagnostic_tests :1872-1913; int16 sums wrap like numpy */
        const double mndwi = (double)wrap16(green - swir1) / (double)wrap16(green + swir1);
        const int16_t mbsrv = wrap16(green + red);
        const int16_t mbsrn = wrap16(nir + swir1);
        const double awesh = (double)blue + 2.5 * (double)green - 1.5 * (double)mbsrn - 0.25 * (double)swir2;
        const double ndvi = (double)wrap16(nir - red) / (double)wrap16(nir + red);
        unsigned dd = 0;
        if (mndwi > p->wigt) dd += 1;
        if (mbsrv > mbsrn) dd += 2;
        if (awesh > p->awgt) dd += 4;
        if (mndwi > p->pswt_1_mndwi && (double)swir1 < p->pswt_1_swir1 && (double)nir < p->pswt_1_nir &&
            ndvi < p->pswt_1_ndvi)
            dd += 8;
        if (mndwi > p->pswt_2_mndwi && (double)blue < p->pswt_2_blue && (double)swir1 < p->pswt_2_swir1 &&
            (double)swir2 < p->pswt_2_swir2 && (double)nir < p->pswt_2_nir)
            dd += 16;
        /* flag_offset_and_scale_inputs (:2300-2302): the same tests on float32 reflectances, every operation a float32
         * operation as numpy performs it on float32 arrays (volatile stores keep each intermediate a rounded float;
         * Python-float thresholds are weak scalars: rounded to float32 first) */
        float nir_f = 0.0f;
        if (p->offset_and_scale_inputs) {
            volatile float f[6], t;
            for (int k = 0; k < 6; ++k) {
                t = (float)v[k] - (float)p->band_offset[k];
                f[k] = (float)p->band_scale[k] * t;
            }
            const float fb = f[0], fg = f[1], fr = f[2], fn = f[3], fs1 = f[4], fs2 = f[5];
            volatile float num = fg - fs1, den = fg + fs1;
            volatile float mndwi_f = num / den;
            volatile float mbsrv_f = fg + fr, mbsrn_f = fn + fs1;
            volatile float a1 = 2.5f * fg, a2 = 1.5f * mbsrn_f, a3 = 0.25f * fs2;
            volatile float aw = fb + a1;
            aw = aw - a2;
            aw = aw - a3;
            volatile float num2 = fn - fr, den2 = fn + fr;
            volatile float ndvi_f = num2 / den2;
            dd = 0;
            if (mndwi_f > (float)p->wigt) dd += 1;
            if (mbsrv_f > mbsrn_f) dd += 2;
            if (aw > (float)p->awgt) dd += 4;
            if (mndwi_f > (float)p->pswt_1_mndwi && fs1 < (float)p->pswt_1_swir1 && fn < (float)p->pswt_1_nir &&
                ndvi_f < (float)p->pswt_1_ndvi)
                dd += 8;
            if (mndwi_f > (float)p->pswt_2_mndwi && fb < (float)p->pswt_2_blue && fs1 < (float)p->pswt_2_swir1 &&
                fs2 < (float)p->pswt_2_swir2 && fn < (float)p->pswt_2_nir)
                dd += 16;
            nir_f = fn;
        }
        const int nir_le_aerosol_max = p->offset_and_scale_inputs ? nir_f <= (float)p->aerosol_max_nir
                                                                  : (double)nir <= p->aerosol_max_nir;
        const int nir_bright = p->offset_and_scale_inputs ? nir_f > (float)p->lcmask_nir : (double)nir > p->lcmask_nir;
        if (invalid) dd = 32; /* :5227 */
        /* generate_interpreted_layer :1687-1707, _get_binary_representation :4286-4317 */
        uint8_t w1 = dd < 32 ? k_interp[dd] : 255;
        uint16_t diag;
        if (dd & 32) diag = 65535;
        else diag = (uint16_t)((dd & 1) + 10 * ((dd >> 1) & 1) + 100 * ((dd >> 2) & 1) +
                               1000 * ((dd >> 3) & 1) + 10000 * ((dd >> 4) & 1));
        const int has_ocean = in->ocean != NULL;
        const uint8_t oc = has_ocean ? in->ocean[i] : 1;
        if (has_ocean && oc == 0) w1 = 254; /* :5245 */
        if (invalid) w1 = 255;              /* :5249 */
        /* _compute_preliminary_cloud_layer :1919-1993 */
        uint8_t cl = 0;
        if (fm & 8) cl = 1;
        if (p->mask_adjacent_to_cloud_mode == DSWX_ADJ_MASK && (fm & 4)) cl = 1;
        if (fm & 2) cl += 4;
        /* coverage :5104-5112 */
        const int valid = !invalid && (!has_ocean || oc != 0);
        n_valid += valid;
        n_cloud += (valid && cl != 0);
        n_not_ocean += has_ocean ? oc : 1;
        /* _apply_aerosol_class_remapping :1249-1302, classes visited 0,2,3,4 */
        uint8_t w1a = w1;
        if (p->apply_aerosol_class_remapping) {
            static const int cls_of_row[4] = {0, 2, 3, 4};
            for (int k = 0; k < 4; ++k) {
                if (p->aerosol_fmask_lut[k][fm] && w1a == cls_of_row[k] && nir_le_aerosol_max) {
                    w1a = 1;
                    if (cl != 255) cl |= 8;
                }
            }
        }
        /* _apply_landcover_and_shadow_masks :1305-1378 */
        uint8_t w2 = w1a;
        const int water = w1a >= 1 && w1a <= 4, psw = w1a == 3 || w1a == 4;
        if (in->shad != NULL) {
            const int land_is_water = in->land != NULL && in->land[i] == 200;
            if (in->shad[i] == 0 && !land_is_water && water) w2 = 0;
        }
        if (in->land != NULL) {
            const uint8_t ld = in->land[i];
            const int bright = nir_bright;
            if (ld == 201 && bright && psw) w2 = 0;
            if (ld < 100 && bright && psw) w2 = 0;
            if (ld >= 100 && ld < 200 && water) w2 = 0;
        }
        /* _add_snow_to_cloud_layer :2052, :2080-2086 */
        if (fm & 16) cl += 2;
        if (w2 == 255) cl = 255;
        /* _apply_cloud_masking :2089-2133 */
        uint8_t w = w2;
        if (cl != 0 && cl != 8) w = 253;
        if (cl == 2 || cl == 10) w = 252;
        if (w2 == 254) w = 254;
        if (w2 == 255) w = 255;
        /* _get_binary_water_layer :1710-1730 */
        const uint8_t bw = (w >= 1 && w <= 4) ? 1 : w;
        /* _get_confidence_layer :1733-1837 */
        uint8_t cf = w2;
        const int cloudy = cl == 1 || cl == 3 || cl == 4 || cl == 5 || cl == 6 || cl == 7 || cl == 9 ||
                           cl == 11 || cl == 12 || cl == 13 || cl == 14 || cl == 15;
        if (cf <= 4 && cloudy) cf = (uint8_t)(cf + 10);
        if (cf <= 4 && cl == 2) cf = (uint8_t)(cf + 20);
        /* _compute_browse_array :3110-3128 on the uncollapsed WTR */
        uint8_t br = w;
        if (p->browse_exclude_psw_aggressive && br == 4) br = 0;
        if (p->collapse_wtr_classes) br = collapse(br);
        if (p->browse_not_water_to_nodata && br == 0) br = 255;
        if (p->browse_cloud_to_nodata && br == 253) br = 255;
        if (p->browse_snow_to_nodata && br == 252) br = 255;
        if (p->browse_ocean_masked_to_nodata && br == 254) br = 255;
        if (p->collapse_wtr_classes) {
            w1 = collapse(w1); w1a = collapse(w1a); w2 = collapse(w2); w = collapse(w);
        }
        if (out->diag) out->diag[i] = diag;
        if (out->wtr1) out->wtr1[i] = w1;
        if (out->wtr1_aerosol) out->wtr1_aerosol[i] = w1a;
        if (out->wtr2) out->wtr2[i] = w2;
        if (out->wtr) out->wtr[i] = w;
        if (out->bwtr) out->bwtr[i] = bw;
        if (out->conf) out->conf[i] = cf;
        if (out->cloud) out->cloud[i] = cl;
        if (out->browse) out->browse[i] = br;
        if (out->mndwi) out->mndwi[i] = mndwi;
        if (out->ndvi) out->ndvi[i] = ndvi;
        if (out->awesh) out->awesh[i] = awesh;
    }
    if (counters) {
        counters[0] += n_valid;
        counters[1] += n_cloud;
        counters[2] += n_not_ocean;
    }
    return 0;
}

/*
 * Enumerates every (n, d) in int16 x int16 with n in [n_lo, n_hi) and counts the
 * pairs where the division-free predicate of the HIP kernel
 *     gt:  (fma(-t, d, n) >  h_up * d) xor (d < 0),  h_up = (nextup(t) - t)/2
 *     lt:  (fma(-t, d, n) < -h_dn * d) xor (d < 0),  h_dn = (t - nextdown(t))/2
 *     (2^-100 replaces the half gap at t == 0, where it is not representable)
 * differs from the reference's  (double)n/(double)d > t  (resp. < t).
 * Returns the mismatch count (expected 0); first mismatch in bad_n/bad_d.
 */
int64_t oracle_check_quotient_predicate(double t, int less_than, int n_lo, int n_hi, int* bad_n,
                                        int* bad_d) {
    /* same constants as make_dev_params() in proteus_amd/csrc/dswx_hip.hip */
    const double h_up = t == 0.0 ? ldexp(1.0, -100) : (nextafter(t, INFINITY) - t) * 0.5;
    const double h_dn = t == 0.0 ? -ldexp(1.0, -100) : -((t - nextafter(t, -INFINITY)) * 0.5);
    int64_t bad = 0;
    for (int n = n_lo; n < n_hi; ++n) {
        const double dn = (double)n;
        for (int d = -32768; d <= 32767; ++d) {
            const double dd = (double)d;
            volatile double q = dn / dd; /* numpy: inf / nan for d == 0 */
            const double r = fma(-t, dd, dn);
            int ref, got;
            if (less_than) {
                ref = q < t;
                got = (r < h_dn * dd) != (d < 0);
            } else {
                ref = q > t;
                got = (r > h_up * dd) != (d < 0);
            }
            if (ref != got) {
                if (bad == 0) { if (bad_n) *bad_n = n; if (bad_d) *bad_d = d; }
                ++bad;
            }
        }
    }
    return bad;
}
