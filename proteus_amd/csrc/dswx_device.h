// dswx_device.h -- device-side types and inline functions shared by every translation unit
// of libdswx_hip.so (production kernels, experimental variants, roofline probes).
// See dswx_hip.hip for the overview and the exactness argument.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "dswx_hip.h"

// ------------------------------------------------------------------------------
// device-side parameter block (derived on the host from dswx_params_t)
// ------------------------------------------------------------------------------
struct DevParams {
    double qt[4];       // 0 wigt, 1 pswt_1_mndwi, 2 pswt_2_mndwi, 3 pswt_1_ndvi
    double qh[4];       // half gap to the neighbouring double (sign folded for [3])
    int32_t awesh4_min;     // 4*AWESH >= this  <=>  awesh > awgt
    int32_t p1_swir1_max;   // x <= max  <=>  x < threshold
    int32_t p1_nir_max;
    int32_t p2_blue_max;
    int32_t p2_swir1_max;
    int32_t p2_swir2_max;
    int32_t p2_nir_max;
    int32_t lc_nir_min;     // nir >= min  <=>  nir > lcmask_nir
    int32_t aer_nir_max;    // nir <= max  <=>  nir <= AEROSOL_REMAPPING_MAX_NIR
    int32_t band_fill[6];   // INT32_MAX = no fill test
    int32_t fmask_fill;     // -1 = no fill test
    int32_t clip_min;       // reflectances are max()ed with this: 1, or -32768 (= no clip)
    int32_t shadow_bits;    // Fmask bits raising CLOUD bit 0: 8, or 8|4 in 'mask' mode
    int32_t collapse;       // 0 / 1, used as a shift count
    uint32_t browse_lut[3]; // byte k: browse value of the k-th uncollapsed WTR code
                            // (0,1,2,3,4,252,253,254,255), _compute_browse_array :3057-3129
    uint32_t aer_lut[64];   // byte v: bit c set <=> Fmask v remaps WTR-1 class c
                            // (all zero when aerosol remapping is disabled)
    // flag_offset_and_scale_inputs (:2300-2302): the chain on float32 reflectances (classify_px_f32, lut_group<.., F32>)
    int32_t f32_mode;
    float f_scale[6], f_offset[6];
    float f_thr[12];        // the twelve thresholds in dswx_params_t order, rounded to float32
    float f_aer_nir;        // AEROSOL_REMAPPING_MAX_NIR rounded to float32
};

struct KArgs {
    DevParams P;
    dswx_planes_in_t in;
    dswx_planes_out_t out;
    uint2* partials;                // fused kernel: per-wave counts, [tile][block][wave] (summed by dswx_counters_finish) ...
    unsigned long long* fold_acc;   // ... or, for launches of a few tiles, accumulators the blocks add to; the block that
    int fold_group_log2;            //   draws a tile's last ticket writes counters[tile] itself (dswx_classify_lut.hip);
                                    //   2^fold_group_log2 = blocks per first-level group
    uint8_t* cover_state;           // 'cover' mode: stage 1 parks one state byte per pixel here
                                    //   (cover_state_of) ...
    uint32_t* cover_bits;           //   ... and the four dilation predicates of every 8-pixel group as one
                                    //   dword of bitmaps (cover_bits_of), [tile][cover_bits_stride];
    long long cover_bits_stride;    //   stage 2 (dswx_cover.hip) reads the bitmaps and leaves
    uint32_t* cover_snow;           //   the final snow decision, one bit per pixel, [tile][cover_snow_stride]
    long long cover_snow_stride;    //   dwords; stage 3 finishes the layers from state + snow
    int height, width;              // 'cover' stage 2 only
    unsigned long long* counters;   // [n_tiles][3] or nullptr
    long long n_pixels;             // per tile
    long long tile_stride;          // pixels between the starts of consecutive tiles in every plane
    long long px_begin;             // generic kernel: first pixel of the tile it covers
    // Ragged contiguous batches (tile_stride % 8 != 0, plane bases 256-byte aligned): every tile starts somewhere inside
    // an 8-pixel group of the planes.  The table-driven kernel then skips the tile's first `head` pixels, head = the
    // distance to the next 8-byte boundary of the u8 planes (= 16-byte boundary of the int16 planes), a number every kernel
    // derives from the tile's address; the generic kernel covers the < 8 head and < 8 tail pixels of every tile.
    int ragged;
    // table-driven kernel: block order.  0 / 1 = a tile's blocks are consecutive in dispatch order (grid = blocks x tiles);
    // G > 1 = the blocks of G tiles interleaved (grid.x = blocks_per_tile * G, grid.y = ceil(tiles / G)), so that the
    // blocks in flight at any moment -- and the 14 streams they read and write -- spread over G tiles of every plane
    int tile_interleave;
    int n_tiles_launch;             // tiles of this launch (interleaved order: the last group may be partial)
    long long blocks_per_tile;      // partials are [tile][blocks_per_tile][wave] whatever the block order
};

// DIAG (5 bits) -> WTR-1 class, interpreted_dswx_band_dict :97-143, as three
// 32-bit masks (bit k of CLS_Bj = bit j of the class of DIAG value k).
static constexpr uint8_t kClassOfDiag[32] = {
    /*00000*/ 0, /*00001*/ 0, /*00010*/ 0, /*00011*/ 4, /*00100*/ 0, /*00101*/ 4,
    /*00110*/ 4, /*00111*/ 2, /*01000*/ 0, /*01001*/ 4, /*01010*/ 4, /*01011*/ 2,
    /*01100*/ 4, /*01101*/ 2, /*01110*/ 2, /*01111*/ 1, /*10000*/ 4, /*10001*/ 4,
    /*10010*/ 4, /*10011*/ 2, /*10100*/ 4, /*10101*/ 2, /*10110*/ 2, /*10111*/ 1,
    /*11000*/ 3, /*11001*/ 2, /*11010*/ 2, /*11011*/ 1, /*11100*/ 2, /*11101*/ 1,
    /*11110*/ 1, /*11111*/ 1};
static constexpr uint32_t class_bit_mask(int bit) {
    uint32_t m = 0;
    for (int k = 0; k < 32; ++k) m |= (uint32_t)((kClassOfDiag[k] >> bit) & 1) << k;
    return m;
}
static constexpr uint32_t CLS_B0 = class_bit_mask(0);
static constexpr uint32_t CLS_B1 = class_bit_mask(1);
static constexpr uint32_t CLS_B2 = class_bit_mask(2);

struct PxOut {
    uint32_t diag, wtr1, wtr1a, wtr2, wtr, bwtr, conf, cloud;
    uint32_t w2_raw, pc;   // uncollapsed WTR-2 and pre-snow CLOUD
    uint32_t state;        // 'cover' stage 1: cover_state_of(w2_raw, pc, snow) (without the adjacent bit)
    uint32_t browse;       // _compute_browse_array of the uncollapsed WTR
};

// fl64(n/d) > t, see the header comment
__device__ __forceinline__ bool quot_gt(double t, double h, double dn, double dd, bool dneg) {
    const double r = __builtin_fma(-t, dd, dn);
    return (r > h * dd) != dneg;
}
// fl64(n/d) < t ; hneg = -(t - nextdown(t))/2
__device__ __forceinline__ bool quot_lt(double t, double hneg, double dn, double dd, bool dneg) {
    const double r = __builtin_fma(-t, dd, dn);
    return (r < hneg * dd) != dneg;
}

__device__ __forceinline__ uint32_t collapse_class(uint32_t v, uint32_t c) {
    // _collapse_wtr_classes :2578-2598 on the value set {0..4, 252..255};
    // c = 1 collapses (0,1,1,2,2), c = 0 is the identity
    return v <= 4u ? (v + c) >> c : v;
}

// 'cover' mode: what stage 2 needs of a pixel, in one byte.  bits 0-2: WTR-2 code (0..4 class, 5 ocean
// masked, 6 fill); bits 3-5: bits 0, 2, 3 of CLOUD before the snow step; bit 6: Fmask snow (bit 4);
// bit 7 (added by the caller): Fmask adjacent-to-cloud (bit 2).
__device__ __forceinline__ uint32_t cover_state_of(uint32_t w2, uint32_t pc, bool snow) {
    const uint32_t code = w2 <= 4u ? w2 : (w2 == 254u ? 5u : 6u);
    return code | (pc & 1u) << 3 | ((pc >> 2) & 3u) << 4 | (snow ? 64u : 0u);
}

// The four predicates of the masked dilations (_add_snow_to_cloud_layer :2055-2078) of one pixel from
// its state byte (adjacent bit included), as bits 0-3: snow seed, area = adjacent & (CLOUD == 0),
// area & (WTR-2 in 1..4), CLOUD == 0.
__device__ __forceinline__ uint32_t cover_flags_of(uint32_t state) {
    const uint32_t clear0 = (state & 0x38u) == 0u ? 1u : 0u;
    const uint32_t area = clear0 & (state >> 7);
    const uint32_t water = ((state & 7u) - 1u) <= 3u ? 1u : 0u;
    return ((state >> 6) & 1u) | area << 1 | (area & water) << 2 | clear0 << 3;
}
// The flags of one pixel spread over a dword, flag q in bit 0 of byte q: pixel j of an 8-pixel group adds
// cover_spread_of(state) << j to the group's bitmap dword (what the table-driven stage 1 looks up per pixel).
__device__ __forceinline__ uint32_t cover_spread_of(uint32_t state) {
    const uint32_t f = cover_flags_of(state);
    return (f & 1u) | ((f >> 1) & 1u) << 8 | ((f >> 2) & 1u) << 16 | ((f >> 3) & 1u) << 24;
}
// Eight pixels' flags -> the bitmap dword stage 1 stores per 8-pixel group: byte q = bit q of the flags
// of pixels 0..7 (pixel j = bit j): [snow8, area8, area-and-water8, clear8].
__device__ __forceinline__ uint32_t cover_bits_of(const uint32_t (&flags)[8]) {
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) m |= ((flags[j] >> q) & 1u) << (8 * q + j);
    return m;
}

// A11-A15 of one pixel, given the uncollapsed WTR-2 class, the CLOUD value before the
// snow bit (A2 + A9) and the snow decision (Fmask bit 4, or the dilated snow mask in
// 'cover' mode).
__device__ __forceinline__ void finish_px(const DevParams& P, uint32_t w2, uint32_t pc, bool snow, PxOut& o) {
    const uint32_t cc = (uint32_t)P.collapse;
    // A11
    uint32_t cl = pc + (snow ? 2u : 0u);
    cl = (w2 == 255u) ? 255u : cl;
    // A12
    uint32_t w = w2;
    w = ((cl != 0u) & (cl != 8u)) ? 253u : w;
    w = ((cl == 2u) | (cl == 10u)) ? 252u : w;
    w = (w2 >= 254u) ? w2 : w;
    // A13
    const uint32_t bw = ((w - 1u) <= 3u) ? 1u : w;
    // A14
    uint32_t cf = w2;
    const bool cloudy = (cl <= 15u) & ((cl & 5u) != 0u);
    cf = ((w2 <= 4u) & cloudy) ? w2 + 10u : cf;
    cf = ((w2 <= 4u) & (cl == 2u)) ? w2 + 20u : cf;
    // A15
    o.wtr2 = collapse_class(w2, cc); o.wtr = collapse_class(w, cc);
    o.bwtr = bw; o.conf = cf; o.cloud = cl;
    // browse layer: nine-entry byte table indexed by the rank of the WTR code
    const uint32_t bi = w <= 4u ? w : 5u + (w & 3u);            // 252..255 -> 5..8
    const uint32_t word = bi < 4u ? P.browse_lut[0] : (bi < 8u ? P.browse_lut[1] : P.browse_lut[2]);
    o.browse = (word >> (8u * (bi & 3u))) & 0xffu;
}

// A5-A8: the five test bits (+ fill, + ocean) -> saved DIAG value and WTR-1 class.
__device__ __forceinline__ void px_w1(uint32_t dd, bool invalid, bool ocean0, uint32_t& diag, uint32_t& w1) {
    // A5-A7: decimal-digit rendering of the bits (:4286-4317), 65535 for fill (:5227)
    const uint32_t digits = (dd & 1u) + 10u * ((dd >> 1) & 1u) + 100u * ((dd >> 2) & 1u) +
                            1000u * ((dd >> 3) & 1u) + 10000u * ((dd >> 4) & 1u);
    diag = invalid ? 65535u : digits;
    const uint32_t cls = ((CLS_B0 >> dd) & 1u) | (((CLS_B1 >> dd) & 1u) << 1) | (((CLS_B2 >> dd) & 1u) << 2);
    // A8
    w1 = ocean0 ? 254u : cls;
    w1 = invalid ? 255u : w1;
}

// A9-A15 once the pixel-dependent predicates are known: `remap` = the Fmask value is in
// the aerosol list of class w1 and nir <= 1000 (:1238-1240), `pc` = preliminary CLOUD
// (A2), `snow` = Fmask bit 4, and the three land-cover / shadow rule hits of :1343-1376.
__device__ __forceinline__ void px_chain(const DevParams& P, uint32_t w1, bool remap, uint32_t pc, bool snow,
                                         bool shadrule, bool lcpsw, bool lchigh, PxOut& o) {
    // A9
    const bool do_remap = remap & (w1 <= 4u);
    const uint32_t w1a = do_remap ? 1u : w1;
    pc |= do_remap ? 8u : 0u;
    // A10 (every predicate reads the input layer; every hit writes 0)
    const bool water = (w1a - 1u) <= 3u;
    const bool psw = (w1a - 3u) <= 1u;
    const bool to_zero = (shadrule & water) | (lcpsw & psw) | (lchigh & water);
    const uint32_t w2 = to_zero ? 0u : w1a;
    const uint32_t cc = (uint32_t)P.collapse;
    o.wtr1 = collapse_class(w1, cc); o.wtr1a = collapse_class(w1a, cc);
    o.w2_raw = w2; o.pc = pc;
    o.state = cover_state_of(w2, pc, snow);
    finish_px(P, w2, pc, snow, o);
}

// One pixel through the whole chain.  b..s2 are the RAW values (sign-extended),
// fm the raw Fmask byte, aer_bits the aerosol table entry of fm (bit c set <=>
// WTR-1 class c is remapped); land/shad/ocean carry neutral sentinels
// (-1 / 1 / 1) when the plane is not given.
__device__ __forceinline__ void classify_px(const DevParams& P, uint32_t aer_bits,
                                            int b, int g, int r, int n, int s1, int s2, int fm,
                                            int land, int shad, int ocean, PxOut& o,
                                            bool& is_valid, bool& is_cloud_and_valid) {
    // A0: cumulative fill test on the raw values, then clip to >= 1
    const bool invalid = (b == P.band_fill[0]) | (g == P.band_fill[1]) | (r == P.band_fill[2]) |
                         (n == P.band_fill[3]) | (s1 == P.band_fill[4]) | (s2 == P.band_fill[5]) |
                         (fm == P.fmask_fill);
    b = max(b, P.clip_min); g = max(g, P.clip_min); r = max(r, P.clip_min);
    n = max(n, P.clip_min); s1 = max(s1, P.clip_min); s2 = max(s2, P.clip_min);
    // A4: int16 wrap-around sums exactly as numpy forms them
    const int d1 = (short)(g + s1), n1 = (short)(g - s1);
    const int mbsrv = (short)(g + r), mbsrn = (short)(n + s1);
    const int n2 = (short)(n - r), d2 = (short)(n + r);
    const double dn1 = (double)n1, dd1 = (double)d1, dn2 = (double)n2, dd2 = (double)d2;
    const bool neg1 = d1 < 0, neg2 = d2 < 0;
    const bool m_wigt = quot_gt(P.qt[0], P.qh[0], dn1, dd1, neg1);
    const bool m_p1 = quot_gt(P.qt[1], P.qh[1], dn1, dd1, neg1);
    const bool m_p2 = quot_gt(P.qt[2], P.qh[2], dn1, dd1, neg1);
    const bool v_p1 = quot_lt(P.qt[3], P.qh[3], dn2, dd2, neg2);
    const int awesh4 = 4 * b + 10 * g - 6 * mbsrn - s2;
    const bool t1 = m_wigt;
    const bool t2 = mbsrv > mbsrn;
    const bool t3 = awesh4 >= P.awesh4_min;
    const bool t4 = m_p1 & (s1 <= P.p1_swir1_max) & (n <= P.p1_nir_max) & v_p1;
    const bool t5 = m_p2 & (b <= P.p2_blue_max) & (s1 <= P.p2_swir1_max) &
                    (s2 <= P.p2_swir2_max) & (n <= P.p2_nir_max);
    const uint32_t dd = (uint32_t)t1 | ((uint32_t)t2 << 1) | ((uint32_t)t3 << 2) |
                        ((uint32_t)t4 << 3) | ((uint32_t)t5 << 4);
    const bool invalid_b = invalid;
    uint32_t w1;
    px_w1(dd, invalid_b, ocean == 0, o.diag, w1);
    // A2
    uint32_t pc = (fm & P.shadow_bits) ? 1u : 0u;
    pc += (fm & 2) ? 4u : 0u;
    // A3 (the counters see the preliminary CLOUD, before the aerosol bit)
    const bool valid = (!invalid) & (ocean != 0);
    is_valid = valid;
    is_cloud_and_valid = valid & (pc != 0u);
    // A9 / A10 predicates that depend on the pixel's own inputs
    const bool remap = (((aer_bits >> (w1 & 7u)) & 1u) != 0u) & (n <= P.aer_nir_max);
    const bool bright = n >= P.lc_nir_min;
    const bool shadrule = (shad == 0) & (land != 200);
    const bool lcpsw = ((land == 201) | ((uint32_t)land < 100u)) & bright;
    const bool lchigh = (uint32_t)(land - 100) < 100u;
    px_chain(P, w1, remap, pc, (fm & 16) != 0, shadrule, lcpsw, lchigh, o);
}

// flag_offset_and_scale_inputs: the same pixel with the reflectances scaled to float32 after fill test and clip
// (:2300-2302: scale_factor * (float32(image) - offset)) and every index, test and nir comparison in float32, operation
// by operation as numpy evaluates :1872-1913, :1238-1240, :1150-1207 on float32 arrays (no contraction: the library is
// built with -ffp-contract=off; hipcc's float32 division is correctly rounded).
__device__ __forceinline__ void classify_px_f32(const DevParams& P, uint32_t aer_bits,
                                                int b, int g, int r, int n, int s1, int s2, int fm,
                                                int land, int shad, int ocean, PxOut& o,
                                                bool& is_valid, bool& is_cloud_and_valid) {
    const bool invalid = (b == P.band_fill[0]) | (g == P.band_fill[1]) | (r == P.band_fill[2]) |
                         (n == P.band_fill[3]) | (s1 == P.band_fill[4]) | (s2 == P.band_fill[5]) |
                         (fm == P.fmask_fill);
    b = max(b, P.clip_min); g = max(g, P.clip_min); r = max(r, P.clip_min);
    n = max(n, P.clip_min); s1 = max(s1, P.clip_min); s2 = max(s2, P.clip_min);
    const float fb = P.f_scale[0] * ((float)b - P.f_offset[0]), fg = P.f_scale[1] * ((float)g - P.f_offset[1]);
    const float fr = P.f_scale[2] * ((float)r - P.f_offset[2]), fn = P.f_scale[3] * ((float)n - P.f_offset[3]);
    const float fs1 = P.f_scale[4] * ((float)s1 - P.f_offset[4]), fs2 = P.f_scale[5] * ((float)s2 - P.f_offset[5]);
    const float mndwi = (fg - fs1) / (fg + fs1);
    const float mbsrv = fg + fr, mbsrn = fn + fs1;
    const float awesh = ((fb + 2.5f * fg) - 1.5f * mbsrn) - 0.25f * fs2;
    const float ndvi = (fn - fr) / (fn + fr);
    const float* T = P.f_thr;       // wigt awgt p1_mndwi p1_nir p1_swir1 p1_ndvi p2_mndwi p2_blue p2_nir p2_swir1 p2_swir2 lcmask_nir
    const bool t1 = mndwi > T[0];
    const bool t2 = mbsrv > mbsrn;
    const bool t3 = awesh > T[1];
    const bool t4 = (mndwi > T[2]) & (fs1 < T[4]) & (fn < T[3]) & (ndvi < T[5]);
    const bool t5 = (mndwi > T[6]) & (fb < T[7]) & (fs1 < T[9]) & (fs2 < T[10]) & (fn < T[8]);
    const uint32_t dd = (uint32_t)t1 | ((uint32_t)t2 << 1) | ((uint32_t)t3 << 2) |
                        ((uint32_t)t4 << 3) | ((uint32_t)t5 << 4);
    uint32_t w1;
    px_w1(dd, invalid, ocean == 0, o.diag, w1);
    uint32_t pc = (fm & P.shadow_bits) ? 1u : 0u;
    pc += (fm & 2) ? 4u : 0u;
    const bool valid = (!invalid) & (ocean != 0);
    is_valid = valid;
    is_cloud_and_valid = valid & (pc != 0u);
    const bool remap = (((aer_bits >> (w1 & 7u)) & 1u) != 0u) & (fn <= P.f_aer_nir);
    const bool bright = fn > T[11];
    const bool shadrule = (shad == 0) & (land != 200);
    const bool lcpsw = ((land == 201) | ((uint32_t)land < 100u)) & bright;
    const bool lchigh = (uint32_t)(land - 100) < 100u;
    px_chain(P, w1, remap, pc, (fm & 16) != 0, shadrule, lcpsw, lchigh, o);
}

// ragged batches (KArgs::ragged): pixels of a tile in front of the first 8-pixel boundary of the planes
__device__ __forceinline__ int ragged_head(const uint8_t* fmask_at_tile_start) {
    return (8 - (int)(reinterpret_cast<uintptr_t>(fmask_at_tile_start) & 7u)) & 7;
}

template <typename T, bool NT> __device__ __forceinline__ T ldg(const void* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const T*>(p));
    return *reinterpret_cast<const T*>(p);
}
template <typename T, bool NT> __device__ __forceinline__ void stg(void* p, T v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<T*>(p));
    else *reinterpret_cast<T*>(p) = v;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// the same vectors at ANY address (int16 planes: 2-byte aligned, byte planes: 1): gfx950 performs unaligned 16- / 8-byte
// global accesses in hardware and the compiler emits the same global_load_dwordx4 / dwordx2 for these types (a wave
// access that straddles cache lines costs a line more, nothing else) -- the direct kernel takes any plane this way
typedef u32x4 __attribute__((aligned(2))) u32x4_u;
typedef u32x2 __attribute__((aligned(1))) u32x2_u;
template <typename TU, typename T, bool NT> __device__ __forceinline__ T ldg_u(const void* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const TU*>(p));
    return *reinterpret_cast<const TU*>(p);
}
template <typename TU, typename T, bool NT> __device__ __forceinline__ void stg_u(void* p, T v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<TU*>(p));
    else *reinterpret_cast<TU*>(p) = v;
}
typedef const __attribute__((address_space(1))) void* gptr_t;     // LDS-DMA source
typedef __attribute__((address_space(3))) void* lptr_t;           // LDS-DMA destination

__device__ __forceinline__ int s16_of(uint32_t dword, int half) {
    return half ? ((int)dword >> 16) : (int)(short)(dword & 0xffffu);
}
__device__ __forceinline__ int u8_of(uint32_t dword, int k) { return (int)((dword >> (8 * k)) & 0xffu); }

// block-wide sum of three per-thread counts -> one atomic per block and counter
__device__ __forceinline__ void reduce_counters(unsigned long long* __restrict__ dst, uint32_t* red,
                                                uint32_t c0, uint32_t c1, uint32_t c2) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_xor(c0, off);
        c1 += __shfl_xor(c1, off);
        c2 += __shfl_xor(c2, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave * 3 + 0] = c0; red[wave * 3 + 1] = c1; red[wave * 3 + 2] = c2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long s = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w * 3 + threadIdx.x];
        if (s) atomicAdd(dst + threadIdx.x, s);
    }
}

