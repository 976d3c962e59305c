// dswx_hip.hip -- MI355X (gfx950 / CDNA4) DSWx-HLS per-pixel classifier.
//
// One fused streaming kernel computes, per pixel, everything the reference does
// in ~100 whole-array numpy passes between src/proteus/dswx_hls.py:5088 and :5369:
//   A0 fill/clip  A2 preliminary CLOUD  A3 coverage counters  A4 five DIAG tests
//   A5-A7 DIAG fill + decimal-digit rendering + WTR-1 LUT  A8 ocean/invalid
//   A9 aerosol remap  A10 land-cover / terrain-shadow  A11 snow -> CLOUD
//   A12 WTR  A13 BWTR  A14 CONF  A15 collapse      (row ids: SURVEY.md §8a)
//
// Roofline: pure HBM streaming, 13 B read + 8 B written per pixel (16 + 8 with
// LAND/SHAD/OCEAN).  No MFMA: there is no contraction anywhere in this path.
//
// Exactness of the float64 threshold tests without a division
// ------------------------------------------------------------
// The reference evaluates  fl64(n/d) > t  with n, d int16 (wrapped sums) and t a
// double (:1872, :1890-1913).  Rounding is monotonic, so
//     fl64(n/d) > t   <=>   n/d > m,   m = (t + nextup(t))/2   (real midpoint)
// and n/d == m is impossible (m has a 54-bit odd significand, n/d has |d| < 2^16).
// With h = (nextup(t) - t)/2 (a power of two, exact in double):
//     d > 0:  n/d > m  <=>  n - t*d > h*d
// r = fma(-t, d, n) is the exact value of n - t*d whenever that needs < 2^53 units
// of the grid both sides live on (always the case near a tie; far from it the sign
// is all that matters and rounding never changes a sign), and h*d is exact.  So
//     fl64(n/d) > t   <=>   (fma(-t,d,n) > h*d)  xor  (d < 0)
// including d == 0 (numpy gives +-inf / nan there and the formula degenerates to
// n > 0).  (At t == 0 the half gap underflows; a stand-in h = 2^-100 is used, see
// make_dev_params.)  `<` is the mirror image with the lower midpoint.  tests/ checks this
// exhaustively over all 2^32 (n, d) pairs against true division.
//
// AWESH (:1881) is a multiple of 0.25 and exact in double, so 4*AWESH is compared
// as an int32.  Integer-vs-double comparisons (:1898-1912, :1361, :1238) become
// integer comparisons against floor/ceil of the threshold, computed on the host.

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cstdarg>
#include <limits>
#include <string>
#include <vector>

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
};

struct KArgs {
    DevParams P;
    dswx_planes_in_t in;
    dswx_planes_out_t out;
    uint2* partials;                // fused kernel: per-wave counts, [tile][block][wave]
    uint8_t* u8_out[7];             // fused kernel: the wanted u8 layers, compacted,
    int u8_region[7];               //   and the LDS staging region each one lives in
    int n_u8_out;
    int n_diag_pieces;              // 8 if DIAG is wanted, else 0
    uint8_t* cover_w2;              // 'cover' mode: stage 1 parks the uncollapsed WTR-2 and
    uint8_t* cover_pc;              //   the pre-snow CLOUD here; stage 2 reads them back
    int height, width;              // 'cover' stage 2 only
    unsigned long long* counters;   // [n_tiles][3] or nullptr
    long long n_pixels;             // per tile
    long long px_begin;             // generic kernel: first pixel of the tile it covers
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
    uint32_t w2_raw, pc;   // uncollapsed WTR-2 and pre-snow CLOUD ('cover' stage 1)
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

// ------------------------------------------------------------------------------
// Fused kernel, direct-store variant (the default): block = 256 threads, one
// 8-pixel group per thread; 16-byte loads from the six int16 planes, 8-byte loads
// from the u8 planes; 16-byte DIAG store and 8-byte u8 stores straight from
// registers, all non-temporal.  grid.y = tile.  Measured 5.1 TB/s (64 tiles); the
// trivial-math probe of the same access shape reaches 5.3 TB/s.
// ------------------------------------------------------------------------------
// EXTRAS: also produce the browse plane and the two scratch planes of 'cover' stage 1
// (kept out of the default instantiation so that its register and instruction budget
// is untouched)
template <bool MASKS, bool EXTRAS, int WPS = 4>
__global__ __launch_bounds__(256, WPS) void dswx_classify_v8(const KArgs a) {
    const DevParams& P = a.P;
    // aerosol table: 256 bytes = one dword per lane of a wave, looked up with
    // ds_bpermute (no LDS storage, no barrier)
    const uint32_t lut_reg = a.P.aer_lut[threadIdx.x & 63];

    const long long n_groups = a.n_pixels >> 3;
    const long long grp = (long long)blockIdx.x * 256 + threadIdx.x;
    // A3: per-wave counts from lane-mask popcounts (scalar unit), no atomics
    uint32_t w_valid = 0, w_cloud = 0, t_ocean = 0;
    // No divergence: threads past the tile's last group redo that group (their
    // results are never stored or counted), so every lane stays active for the
    // cross-lane table lookup below.
    const bool in_range = grp < n_groups;
    {
        const long long off = (long long)blockIdx.y * a.n_pixels + (in_range ? grp : n_groups - 1) * 8;
        u32x4 v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = ldg<u32x4, true>(a.in.band[k] + off);
        const u32x2 vf = ldg<u32x2, true>(a.in.fmask + off);
        u32x2 vl = {0u, 0u}, vs = {0u, 0u}, vo = {0u, 0u};
        bool has_l = false, has_s = false, has_o = false;
        if (MASKS) {
            has_l = a.in.land != nullptr; has_s = a.in.shad != nullptr; has_o = a.in.ocean != nullptr;
            if (has_l) vl = ldg<u32x2, true>(a.in.land + off);
            if (has_s) vs = ldg<u32x2, true>(a.in.shad + off);
            if (has_o) {
                vo = ldg<u32x2, true>(a.in.ocean + off);
                t_ocean = __builtin_amdgcn_sad_u8(vo.x, 0u, __builtin_amdgcn_sad_u8(vo.y, 0u, 0u));
                t_ocean = in_range ? t_ocean : 0u;
            }
        }
        uint32_t q_diag[4] = {0, 0, 0, 0};
        uint32_t q_w1[2] = {0, 0}, q_w1a[2] = {0, 0}, q_w2[2] = {0, 0}, q_w[2] = {0, 0},
                 q_bw[2] = {0, 0}, q_cf[2] = {0, 0}, q_cl[2] = {0, 0}, q_w2r[2] = {0, 0}, q_pc[2] = {0, 0}, q_br[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int wd = j >> 1, hf = j & 1;
            const int b = s16_of(v[0][wd], hf), g = s16_of(v[1][wd], hf), r = s16_of(v[2][wd], hf),
                      n = s16_of(v[3][wd], hf), s1 = s16_of(v[4][wd], hf), s2 = s16_of(v[5][wd], hf);
            const int bw = j >> 2, bk = j & 3;
            const int fm = u8_of(vf[bw], bk);
            int land = -1, shad = 1, ocean = 1;
            if (MASKS) {
                if (has_l) land = u8_of(vl[bw], bk);
                if (has_s) shad = u8_of(vs[bw], bk);
                if (has_o) ocean = u8_of(vo[bw], bk);
            }
            const uint32_t aer_bits =
                ((uint32_t)__builtin_amdgcn_ds_bpermute((fm >> 2) << 2, (int)lut_reg) >> (8 * (fm & 3))) & 0xffu;
            PxOut o;
            bool ok, cv;
            classify_px(P, aer_bits, b, g, r, n, s1, s2, fm, land, shad, ocean, o, ok, cv);
            w_valid += (uint32_t)__popcll(__ballot(ok & in_range));
            w_cloud += (uint32_t)__popcll(__ballot(cv & in_range));
            q_diag[wd] |= o.diag << (16 * hf);
            q_w1[bw] |= o.wtr1 << (8 * bk);
            q_w1a[bw] |= o.wtr1a << (8 * bk);
            q_w2[bw] |= o.wtr2 << (8 * bk);
            q_w[bw] |= o.wtr << (8 * bk);
            q_bw[bw] |= o.bwtr << (8 * bk);
            q_cf[bw] |= o.conf << (8 * bk);
            q_cl[bw] |= o.cloud << (8 * bk);
            if (EXTRAS) {
                q_w2r[bw] |= o.w2_raw << (8 * bk);
                q_pc[bw] |= o.pc << (8 * bk);
                q_br[bw] |= o.browse << (8 * bk);
            }
        }
        if (in_range) {
        if (a.out.diag) stg<u32x4, true>(a.out.diag + off, u32x4{q_diag[0], q_diag[1], q_diag[2], q_diag[3]});
        if (a.out.wtr1) stg<u32x2, true>(a.out.wtr1 + off, u32x2{q_w1[0], q_w1[1]});
        if (a.out.wtr1_aerosol) stg<u32x2, true>(a.out.wtr1_aerosol + off, u32x2{q_w1a[0], q_w1a[1]});
        if (a.out.wtr2) stg<u32x2, true>(a.out.wtr2 + off, u32x2{q_w2[0], q_w2[1]});
        if (a.out.wtr) stg<u32x2, true>(a.out.wtr + off, u32x2{q_w[0], q_w[1]});
        if (a.out.bwtr) stg<u32x2, true>(a.out.bwtr + off, u32x2{q_bw[0], q_bw[1]});
        if (a.out.conf) stg<u32x2, true>(a.out.conf + off, u32x2{q_cf[0], q_cf[1]});
        if (a.out.cloud) stg<u32x2, true>(a.out.cloud + off, u32x2{q_cl[0], q_cl[1]});
        if (EXTRAS && a.out.browse) stg<u32x2, true>(a.out.browse + off, u32x2{q_br[0], q_br[1]});
        if (EXTRAS && a.cover_w2) {   // 'cover' stage 1 (wave-uniform)
            *reinterpret_cast<u32x2*>(a.cover_w2 + off) = u32x2{q_w2r[0], q_w2r[1]};
            *reinterpret_cast<u32x2*>(a.cover_pc + off) = u32x2{q_pc[0], q_pc[1]};
        }
        }
    }
    if (a.partials) {
        if (MASKS && a.in.ocean != nullptr) {
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) t_ocean += __shfl_xor(t_ocean, sh);
        }
        if ((threadIdx.x & 63) == 0) {
            const long long slot = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
            a.partials[slot] = make_uint2(w_valid | (w_cloud << 16), t_ocean);
        }
    }
}

// ------------------------------------------------------------------------------
// Fused kernel, LDS-staged variant (DSWX_FUSED_VARIANT=1).  Block = 512 threads = 4096 consecutive pixels of one tile
// (grid.y = tile); each thread classifies one 8-pixel group.
//
// Loads: straight to registers, 16 B per lane from each int16 plane and 8 B per
// lane from each u8 plane, non-temporal.  Seven-plane READS stream at the full
// HBM rate in this shape (6.3-7.0 TB/s measured), so they are not staged.
//
// Stores: transposed through LDS.  Measured on MI355X, a wave that scatters
// 512 B - 1 KiB to each of the seven output planes gets 3.9-4.5 TB/s of write
// bandwidth, while a wave that writes one plane in multi-KiB contiguous runs of
// 16-byte stores gets 6.4 TB/s.  So every thread parks its results in LDS
// (36 KiB per block), and after one barrier each of the 8 waves streams whole
// 1 KiB pieces of consecutive plane segments (4 KiB per u8 plane, 8 KiB for DIAG)
// with 16-byte non-temporal stores.
// ------------------------------------------------------------------------------
constexpr int FUSED_THREADS = 512;
constexpr int FUSED_PX = FUSED_THREADS * 8;            // pixels per block
constexpr int STAGE_DIAG_BYTES = FUSED_PX * 2;          // 8 KiB
constexpr int STAGE_U8_BYTES = FUSED_PX;                // 4 KiB per u8 plane
constexpr int STAGE_BYTES = STAGE_DIAG_BYTES + 7 * STAGE_U8_BYTES;

template <bool MASKS>
__global__ __launch_bounds__(FUSED_THREADS) void dswx_classify_fused(const KArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[STAGE_BYTES];
    const DevParams& P = a.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // aerosol table: 256 bytes = one dword per lane of a wave, looked up with
    // ds_bpermute (no LDS storage, no barrier)
    const uint32_t lut_reg = a.P.aer_lut[lane];

    const long long n_groups = a.n_pixels >> 3;
    const long long grp = (long long)blockIdx.x * FUSED_THREADS + threadIdx.x;
    const long long tile_base = (long long)blockIdx.y * a.n_pixels;
    // A3: per-wave counts from lane-mask popcounts (scalar unit), no atomics
    uint32_t w_valid = 0, w_cloud = 0, t_ocean = 0;
    // No divergence: threads past the tile's last group redo that group (their
    // results are never stored or counted), so every lane stays active for the
    // cross-lane table lookup below.
    const bool in_range = grp < n_groups;
    {
        const long long off = tile_base + (in_range ? grp : n_groups - 1) * 8;
        u32x4 v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = ldg<u32x4, true>(a.in.band[k] + off);
        const u32x2 vf = ldg<u32x2, true>(a.in.fmask + off);
        u32x2 vl = {0u, 0u}, vs = {0u, 0u}, vo = {0u, 0u};
        bool has_l = false, has_s = false, has_o = false;
        if (MASKS) {
            has_l = a.in.land != nullptr; has_s = a.in.shad != nullptr; has_o = a.in.ocean != nullptr;
            if (has_l) vl = ldg<u32x2, true>(a.in.land + off);
            if (has_s) vs = ldg<u32x2, true>(a.in.shad + off);
            if (has_o) {
                vo = ldg<u32x2, true>(a.in.ocean + off);
                t_ocean = __builtin_amdgcn_sad_u8(vo.x, 0u, __builtin_amdgcn_sad_u8(vo.y, 0u, 0u));
                t_ocean = in_range ? t_ocean : 0u;
            }
        }
        uint32_t q_diag[4] = {0, 0, 0, 0};
        uint32_t q_w1[2] = {0, 0}, q_w1a[2] = {0, 0}, q_w2[2] = {0, 0}, q_w[2] = {0, 0},
                 q_bw[2] = {0, 0}, q_cf[2] = {0, 0}, q_cl[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int wd = j >> 1, hf = j & 1;
            const int b = s16_of(v[0][wd], hf), g = s16_of(v[1][wd], hf), r = s16_of(v[2][wd], hf),
                      n = s16_of(v[3][wd], hf), s1 = s16_of(v[4][wd], hf), s2 = s16_of(v[5][wd], hf);
            const int bw = j >> 2, bk = j & 3;
            const int fm = u8_of(vf[bw], bk);
            int land = -1, shad = 1, ocean = 1;
            if (MASKS) {
                if (has_l) land = u8_of(vl[bw], bk);
                if (has_s) shad = u8_of(vs[bw], bk);
                if (has_o) ocean = u8_of(vo[bw], bk);
            }
            const uint32_t aer_bits =
                ((uint32_t)__builtin_amdgcn_ds_bpermute((fm >> 2) << 2, (int)lut_reg) >> (8 * (fm & 3))) & 0xffu;
            PxOut o;
            bool ok, cv;
            classify_px(P, aer_bits, b, g, r, n, s1, s2, fm, land, shad, ocean, o, ok, cv);
            w_valid += (uint32_t)__popcll(__ballot(ok & in_range));
            w_cloud += (uint32_t)__popcll(__ballot(cv & in_range));
            q_diag[wd] |= o.diag << (16 * hf);
            q_w1[bw] |= o.wtr1 << (8 * bk);
            q_w1a[bw] |= o.wtr1a << (8 * bk);
            q_w2[bw] |= o.wtr2 << (8 * bk);
            q_w[bw] |= o.wtr << (8 * bk);
            q_bw[bw] |= o.bwtr << (8 * bk);
            q_cf[bw] |= o.conf << (8 * bk);
            q_cl[bw] |= o.cloud << (8 * bk);
        }
        // park the results: region 0 = DIAG (16 B per thread), regions 1..7 = the u8
        // layers in dswx_planes_out_t order (8 B per thread)
        *reinterpret_cast<u32x4*>(stage + threadIdx.x * 16) = u32x4{q_diag[0], q_diag[1], q_diag[2], q_diag[3]};
        uint8_t* su8 = stage + STAGE_DIAG_BYTES + threadIdx.x * 8;
        *reinterpret_cast<u32x2*>(su8 + 0 * STAGE_U8_BYTES) = u32x2{q_w1[0], q_w1[1]};
        if (a.out.wtr1_aerosol) *reinterpret_cast<u32x2*>(su8 + 1 * STAGE_U8_BYTES) = u32x2{q_w1a[0], q_w1a[1]};
        *reinterpret_cast<u32x2*>(su8 + 2 * STAGE_U8_BYTES) = u32x2{q_w2[0], q_w2[1]};
        *reinterpret_cast<u32x2*>(su8 + 3 * STAGE_U8_BYTES) = u32x2{q_w[0], q_w[1]};
        *reinterpret_cast<u32x2*>(su8 + 4 * STAGE_U8_BYTES) = u32x2{q_bw[0], q_bw[1]};
        *reinterpret_cast<u32x2*>(su8 + 5 * STAGE_U8_BYTES) = u32x2{q_cf[0], q_cf[1]};
        *reinterpret_cast<u32x2*>(su8 + 6 * STAGE_U8_BYTES) = u32x2{q_cl[0], q_cl[1]};
    }
    if (a.partials) {
        if (MASKS && a.in.ocean != nullptr) {
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) t_ocean += __shfl_xor(t_ocean, sh);
        }
        if (lane == 0) {
            const long long slot = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (FUSED_THREADS / 64) + wave;
            a.partials[slot] = make_uint2(w_valid | (w_cloud << 16), t_ocean);
        }
    }
    __syncthreads();

    // Store phase.  The block's output is a list of 1 KiB pieces: 8 for DIAG (if
    // wanted), 4 for each wanted u8 layer, in plane order; wave w takes the w-th
    // run of consecutive pieces, so it writes 4 KiB runs of a single plane.
    const long long px0 = (long long)blockIdx.x * FUSED_PX;          // first pixel of the block in its tile
    const long long n_vec = n_groups * 8;                            // pixels the vector path covers
    const int n_pieces = a.n_diag_pieces + 4 * a.n_u8_out;
    const int per_wave = (n_pieces + (FUSED_THREADS / 64) - 1) / (FUSED_THREADS / 64);
    for (int q = 0; q < per_wave; ++q) {
        const int piece = wave * per_wave + q;
        if (piece >= n_pieces) break;
        if (piece < a.n_diag_pieces) {
            const long long p = px0 + piece * 512 + lane * 8;        // 8 px = 16 B of DIAG
            if (p + 8 <= n_vec)
                stg<u32x4, true>(a.out.diag + tile_base + p,
                                 *reinterpret_cast<const u32x4*>(stage + piece * 1024 + lane * 16));
        } else {
            const int u = (piece - a.n_diag_pieces) >> 2, sub = (piece - a.n_diag_pieces) & 3;
            const int region = a.u8_region[u];
            uint8_t* dst = a.u8_out[u] + tile_base;
            const long long p = px0 + sub * 1024 + lane * 16;        // 16 px = 16 B
            const uint8_t* src = stage + STAGE_DIAG_BYTES + region * STAGE_U8_BYTES + sub * 1024 + lane * 16;
            if (p + 16 <= n_vec) stg<u32x4, true>(dst + p, *reinterpret_cast<const u32x4*>(src));
            else if (p + 8 <= n_vec) stg<u32x2, true>(dst + p, *reinterpret_cast<const u32x2*>(src));
        }
    }
}

// ------------------------------------------------------------------------------
// Fused kernel, warp-specialised variant (DSWX_FUSED_VARIANT=2).  Block = 256 threads =
// one 2048-pixel chunk of a tile (grid.y = tile).
//
//  phase A  the chunk's input planes are pulled into LDS with LDS-DMA
//           (global_load_lds, 16 B per lane, 1 KiB per wave-instruction, no VGPRs).
//           The 1 KiB pieces are dealt to the waves in plane order, 8 consecutive
//           pieces each: waves 0-2 read two whole 4 KiB band segments, wave 3 the u8
//           planes -- every wave streams whole contiguous plane segments;
//  phase B  each thread reads its 8 pixels from the LDS images (ds_read_b128 / b64),
//           barrier, classifies them exactly as the direct kernel does, and parks the
//           results in LDS *over* the input images (they are dead by then);
//  phase C  each wave writes 4 consecutive 1 KiB pieces of the output planes with
//           16-byte non-temporal stores (4 KiB DIAG runs, 2 KiB u8 runs).
// LDS per block: 26 KiB (32 KiB with LAND/SHAD/OCEAN) -> 5-6 blocks per CU.
// The trivial-math probe of this data movement (dswx_ws_probe_k) runs ~15 % above the
// direct-store probe on the same device.
// ------------------------------------------------------------------------------
constexpr int WS_PX = 2048;
constexpr int WS_BAND_BYTES = WS_PX * 2, WS_U8_BYTES = WS_PX;
constexpr int WS_IN_FMASK = 6 * WS_BAND_BYTES;                    // 24 KiB
constexpr int WS_IN_MASKS = WS_IN_FMASK + WS_U8_BYTES;            // land, shad, ocean follow
constexpr int WS_OUT_U8 = WS_BAND_BYTES;                          // after the 4 KiB DIAG image

template <bool MASKS>
__global__ __launch_bounds__(256, MASKS ? 4 : 5) void dswx_classify_ws(const KArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[WS_IN_MASKS + (MASKS ? 3 * WS_U8_BYTES : 0)];
    const DevParams& P = a.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lut_reg = a.P.aer_lut[lane];
    const long long tile_base = (long long)blockIdx.y * a.n_pixels;
    const long long px0 = (long long)blockIdx.x * WS_PX;
    const long long n_vec = (a.n_pixels >> 3) << 3;       // pixels the vector path covers
    // last byte offsets a 16-byte access may start at without leaving the covered range
    // (n_vec >= 8; a shorter final access re-reads in-range bytes, never stored or counted)
    const long long last16_i16 = (n_vec - 8) * 2, last16_u8 = n_vec >= 16 ? n_vec - 16 : 0;

    // ---- phase A: LDS-DMA, pieces of 1 KiB in plane order
    const bool has_l = MASKS && a.in.land, has_s = MASKS && a.in.shad, has_o = MASKS && a.in.ocean;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int piece = wave * 8 + q;
        if (piece < 24) {
            const int plane = piece >> 2, sub = piece & 3;
            long long byte = (px0 * 2) + sub * 1024 + lane * 16;
            byte = byte <= last16_i16 ? byte : last16_i16;
            __builtin_amdgcn_global_load_lds(
                (gptr_t)(reinterpret_cast<const uint8_t*>(a.in.band[plane]) + tile_base * 2 + byte),
                (lptr_t)(lds + plane * WS_BAND_BYTES + sub * 1024), 16, 0, 2);
        } else {
            const int u = (piece - 24) >> 1, sub = (piece - 24) & 1;      // 0 fmask, 1 land, 2 shad, 3 ocean
            const uint8_t* src = u == 0 ? a.in.fmask : (u == 1 ? a.in.land : (u == 2 ? a.in.shad : a.in.ocean));
            const bool present = u == 0 || (MASKS && ((u == 1 && has_l) || (u == 2 && has_s) || (u == 3 && has_o)));
            if (present) {
                long long byte = px0 + sub * 1024 + lane * 16;
                byte = byte <= last16_u8 ? byte : last16_u8;
                __builtin_amdgcn_global_load_lds((gptr_t)(src + tile_base + byte),
                                                 (lptr_t)(lds + WS_IN_FMASK + u * WS_U8_BYTES + sub * 1024), 16, 0, 2);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- phase B: registers <- LDS images
    const long long grp = (px0 >> 3) + threadIdx.x;
    const bool in_range = grp < (a.n_pixels >> 3);
    u32x4 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = *reinterpret_cast<const u32x4*>(lds + k * WS_BAND_BYTES + threadIdx.x * 16);
    const u32x2 vf = *reinterpret_cast<const u32x2*>(lds + WS_IN_FMASK + threadIdx.x * 8);
    u32x2 vl = {0u, 0u}, vs = {0u, 0u}, vo = {0u, 0u};
    uint32_t w_valid = 0, w_cloud = 0, t_ocean = 0;
    if (MASKS) {
        if (has_l) vl = *reinterpret_cast<const u32x2*>(lds + WS_IN_MASKS + threadIdx.x * 8);
        if (has_s) vs = *reinterpret_cast<const u32x2*>(lds + WS_IN_MASKS + WS_U8_BYTES + threadIdx.x * 8);
        if (has_o) {
            vo = *reinterpret_cast<const u32x2*>(lds + WS_IN_MASKS + 2 * WS_U8_BYTES + threadIdx.x * 8);
            t_ocean = __builtin_amdgcn_sad_u8(vo.x, 0u, __builtin_amdgcn_sad_u8(vo.y, 0u, 0u));
            t_ocean = in_range ? t_ocean : 0u;
        }
    }
    __syncthreads();                                     // the input images are dead from here on

    uint32_t q_diag[4] = {0, 0, 0, 0};
    uint32_t q_w1[2] = {0, 0}, q_w1a[2] = {0, 0}, q_w2[2] = {0, 0}, q_w[2] = {0, 0},
             q_bw[2] = {0, 0}, q_cf[2] = {0, 0}, q_cl[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int wd = j >> 1, hf = j & 1;
        const int b = s16_of(v[0][wd], hf), g = s16_of(v[1][wd], hf), r = s16_of(v[2][wd], hf),
                  n = s16_of(v[3][wd], hf), s1 = s16_of(v[4][wd], hf), s2 = s16_of(v[5][wd], hf);
        const int bw = j >> 2, bk = j & 3;
        const int fm = u8_of(vf[bw], bk);
        int land = -1, shad = 1, ocean = 1;
        if (MASKS) {
            if (has_l) land = u8_of(vl[bw], bk);
            if (has_s) shad = u8_of(vs[bw], bk);
            if (has_o) ocean = u8_of(vo[bw], bk);
        }
        const uint32_t aer_bits =
            ((uint32_t)__builtin_amdgcn_ds_bpermute((fm >> 2) << 2, (int)lut_reg) >> (8 * (fm & 3))) & 0xffu;
        PxOut o;
        bool ok, cv;
        classify_px(P, aer_bits, b, g, r, n, s1, s2, fm, land, shad, ocean, o, ok, cv);
        w_valid += (uint32_t)__popcll(__ballot(ok & in_range));
        w_cloud += (uint32_t)__popcll(__ballot(cv & in_range));
        q_diag[wd] |= o.diag << (16 * hf);
        q_w1[bw] |= o.wtr1 << (8 * bk);
        q_w1a[bw] |= o.wtr1a << (8 * bk);
        q_w2[bw] |= o.wtr2 << (8 * bk);
        q_w[bw] |= o.wtr << (8 * bk);
        q_bw[bw] |= o.bwtr << (8 * bk);
        q_cf[bw] |= o.conf << (8 * bk);
        q_cl[bw] |= o.cloud << (8 * bk);
    }
    // park the results over the dead input images: DIAG 4 KiB, then 7 u8 regions of 2 KiB
    *reinterpret_cast<u32x4*>(lds + threadIdx.x * 16) = u32x4{q_diag[0], q_diag[1], q_diag[2], q_diag[3]};
    uint8_t* su8 = lds + WS_OUT_U8 + threadIdx.x * 8;
    *reinterpret_cast<u32x2*>(su8 + 0 * WS_U8_BYTES) = u32x2{q_w1[0], q_w1[1]};
    if (a.out.wtr1_aerosol) *reinterpret_cast<u32x2*>(su8 + 1 * WS_U8_BYTES) = u32x2{q_w1a[0], q_w1a[1]};
    *reinterpret_cast<u32x2*>(su8 + 2 * WS_U8_BYTES) = u32x2{q_w2[0], q_w2[1]};
    *reinterpret_cast<u32x2*>(su8 + 3 * WS_U8_BYTES) = u32x2{q_w[0], q_w[1]};
    *reinterpret_cast<u32x2*>(su8 + 4 * WS_U8_BYTES) = u32x2{q_bw[0], q_bw[1]};
    *reinterpret_cast<u32x2*>(su8 + 5 * WS_U8_BYTES) = u32x2{q_cf[0], q_cf[1]};
    *reinterpret_cast<u32x2*>(su8 + 6 * WS_U8_BYTES) = u32x2{q_cl[0], q_cl[1]};
    if (a.partials) {
        if (MASKS && has_o) {
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) t_ocean += __shfl_xor(t_ocean, sh);
        }
        if (lane == 0) {
            const long long slot = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
            a.partials[slot] = make_uint2(w_valid | (w_cloud << 16), t_ocean);
        }
    }
    __syncthreads();

    // ---- phase C: 1 KiB pieces in plane order (DIAG 4, each wanted u8 layer 2), consecutive per wave
    const int n_pieces = (a.n_diag_pieces ? 4 : 0) + 2 * a.n_u8_out;
    const int per_wave = (n_pieces + 3) / 4;
    const int diag_pieces = a.n_diag_pieces ? 4 : 0;
    for (int q = 0; q < per_wave; ++q) {
        const int piece = wave * per_wave + q;
        if (piece >= n_pieces) break;
        if (piece < diag_pieces) {
            const long long p = px0 + piece * 512 + lane * 8;
            if (p + 8 <= n_vec)
                stg<u32x4, true>(a.out.diag + tile_base + p, *reinterpret_cast<const u32x4*>(lds + piece * 1024 + lane * 16));
        } else {
            const int u = (piece - diag_pieces) >> 1, sub = (piece - diag_pieces) & 1;
            const int region = a.u8_region[u];
            uint8_t* dst = a.u8_out[u] + tile_base;
            const long long p = px0 + sub * 1024 + lane * 16;
            const uint8_t* src = lds + WS_OUT_U8 + region * WS_U8_BYTES + sub * 1024 + lane * 16;
            if (p + 16 <= n_vec) stg<u32x4, true>(dst + p, *reinterpret_cast<const u32x4*>(src));
            else if (p + 8 <= n_vec) stg<u32x2, true>(dst + p, *reinterpret_cast<const u32x2*>(src));
        }
    }
}

// ==============================================================================
// Table-driven fused kernel (DSWX_FUSED_VARIANT=3)
// ==============================================================================
// The per-pixel chain after the five tests is a pure function of a few bits, so it is
// tabulated ON THE DEVICE by dswx_build_tables -- which calls the very same px_w1 /
// px_chain the scalar path uses (one source of truth) -- and the hot kernel only
//   * does the arithmetic part in packed int16 (v_pk_*: two pixels per instruction) and
//     in sign-bit form (no compare -> lane-mask -> select chains, hence almost no SALU),
//   * looks three small LDS tables up per pixel,
//   * transposes the table words into plane order with v_perm_b32.
struct Tables {
    uint32_t lut1[128];    // [T1 | T2<<1 | !T3<<2 | T4<<3 | T5<<4 | invalid<<5 | ocean0<<6]
                           //   -> DIAG(16) | WTR-1 code(8) << 16 | WTR-1 as saved(8) << 24
    uint16_t fm16[256];    // Fmask byte -> aerosol class bits(5) | shadow<<5 | cloud<<6 | snow<<7
                           //   | is_fill<<8 | prelim_cloud_nonzero<<9
    uint8_t land8[256];    // LAND byte -> is_water(200) | psw_rule_class(201 or <100)<<1 | high_dev<<2
    uint2 chain[1024];     // [code | remap<<3 | shadow<<4 | cloud<<5 | snow<<6 | shadrule<<7 |
                           //  lcpsw<<8 | lchigh<<9] -> x = WTR-1-AEROSOL | WTR-2<<8 | WTR<<16 | BWTR<<24,
                           //                           y = CONF | CLOUD<<8
};
// WTR-1 "code": 0..4 = class, 5 = ocean masked (254), 6 = fill (255)

__global__ __launch_bounds__(256) void dswx_build_tables(const DevParams P, Tables* __restrict__ t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t cc = (uint32_t)P.collapse;
    if (i < 128) {
        const uint32_t dd = (uint32_t)(i & 3) | ((((uint32_t)i >> 2) & 1u) ^ 1u) << 2 | (((uint32_t)i >> 3) & 3u) << 3;
        uint32_t diag, w1;
        px_w1(dd, (i >> 5) & 1, (i >> 6) & 1, diag, w1);
        const uint32_t code = w1 <= 4u ? w1 : (w1 == 254u ? 5u : 6u);
        t->lut1[i] = diag | (code << 16) | (collapse_class(w1, cc) << 24);
    }
    if (i < 256) {
        const uint32_t aer = (P.aer_lut[i >> 2] >> (8 * (i & 3))) & 0x1fu;
        const uint32_t shadow = (i & P.shadow_bits) ? 1u : 0u, cloud = (i >> 1) & 1u, snow = (i >> 4) & 1u;
        t->fm16[i] = (uint16_t)(aer | shadow << 5 | cloud << 6 | snow << 7 | (i == P.fmask_fill ? 1u : 0u) << 8 |
                                (shadow | cloud) << 9);
        t->land8[i] = (uint8_t)((i == 200 ? 1 : 0) | ((i == 201 || i < 100) ? 2 : 0) | ((i >= 100 && i < 200) ? 4 : 0));
    }
    if (i < 1024) {
        const uint32_t code = i & 7;
        const uint32_t w1 = code <= 4u ? code : (code == 5u ? 254u : 255u);
        const uint32_t pc = ((i >> 4) & 1u) + 4u * ((i >> 5) & 1u);
        PxOut o;
        px_chain(P, w1, (i >> 3) & 1, pc, (i >> 6) & 1, (i >> 7) & 1, (i >> 8) & 1, (i >> 9) & 1, o);
        t->chain[i] = make_uint2(o.wtr1a | o.wtr2 << 8 | o.wtr << 16 | o.bwtr << 24, o.conf | o.cloud << 8);
    }
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b));
}
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) {   // signed, saturating
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max_i(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t hi32(double x) { return (uint32_t)(__builtin_bit_cast(unsigned long long, x) >> 32); }
__device__ __forceinline__ uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }

// packed / derived constants of one launch (kernel argument)
struct LutConsts {
    uint32_t fill_pk[6], fill_off[6];   // x ^ fill_pk, | fill_off (0x00010001 disables a band's test)
    uint32_t clip_pk;                   // packed clip minimum
    uint32_t k_p1_swir1, k_p1_nir, k_p2_blue, k_p2_swir1, k_p2_swir2, k_p2_nir, k_lc_nir, k_aer_nir;   // packed
    uint32_t force4, force5, force_dark, force_noaer;   // 0x80008000 when a threshold lies outside int16
    int32_t awesh_init;                 // -awesh4_min
};

// 4 pixels' table words -> 4 plane dwords (byte k of every word -> plane k)
__device__ __forceinline__ void transpose4(const uint32_t a[4], uint32_t out[4]) {
    const uint32_t t01l = perm_b32(a[1], a[0], 0x05010400u), t01h = perm_b32(a[1], a[0], 0x07030602u);
    const uint32_t t23l = perm_b32(a[3], a[2], 0x05010400u), t23h = perm_b32(a[3], a[2], 0x07030602u);
    out[0] = perm_b32(t23l, t01l, 0x05040100u); out[1] = perm_b32(t23l, t01l, 0x07060302u);
    out[2] = perm_b32(t23h, t01h, 0x05040100u); out[3] = perm_b32(t23h, t01h, 0x07060302u);
}

// The table-driven classification of one 8-pixel group held in registers.  Leaves, per
// pixel j, the three table words (w1w: DIAG | code | WTR-1; chx: WTR-1-AEROSOL, WTR-2, WTR,
// BWTR; chy: CONF, CLOUD) and adds the group's coverage counts to `cnt`.
template <bool MASKS>
__device__ __forceinline__ void lut_group(const DevParams& P, const LutConsts& C, const uint32_t* __restrict__ s_lut1,
                                          const uint16_t* __restrict__ s_fm16, const uint8_t* __restrict__ s_land8,
                                          const uint2* __restrict__ s_chain, const u32x4 (&v)[6], const u32x2 vf,
                                          const u32x2 vl, const u32x2 vs, const u32x2 vo, bool has_l, bool in_range,
                                          uint32_t (&w1w)[8], uint32_t (&chx)[8], uint32_t (&chy)[8], uint32_t& cnt) {
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            // ---- two pixels at a time, packed int16
            uint32_t x[6], e[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { x[k] = v[k][wd]; e[k] = (x[k] ^ C.fill_pk[k]) | C.fill_off[k]; }
            const uint32_t emin = pk_min_u(pk_min_u(pk_min_u(e[0], e[1]), pk_min_u(e[2], e[3])), pk_min_u(e[4], e[5]));
            const uint32_t bandvalid = pk_min_u(emin, 0x00010001u);         // 1 per half: no band equals its fill
#pragma unroll
            for (int k = 0; k < 6; ++k) x[k] = pk_max_i(x[k], C.clip_pk);   // A0 clip
            const uint32_t b = x[0], g = x[1], r = x[2], n = x[3], s1 = x[4], s2 = x[5];
            const uint32_t d1 = pk_add(g, s1), n1 = pk_sub(g, s1), mv = pk_add(g, r), mn = pk_add(n, s1);
            const uint32_t n2 = pk_sub(n, r), d2 = pk_add(n, r);
            // sign bit (15 / 31) set  <=>  ...
            const uint32_t t2s = pk_sub_sat(mn, mv);                                            // T2 true
            const uint32_t viol4 = pk_sub_sat(C.k_p1_swir1, s1) | pk_sub_sat(C.k_p1_nir, n) | C.force4;   // T4 ints fail
            const uint32_t viol5 = pk_sub_sat(C.k_p2_blue, b) | pk_sub_sat(C.k_p2_swir1, s1) |
                                   pk_sub_sat(C.k_p2_swir2, s2) | pk_sub_sat(C.k_p2_nir, n) | C.force5;   // T5 ints fail
            const uint32_t dark = pk_sub_sat(n, C.k_lc_nir) | C.force_dark;                     // nir NOT > lcmask_nir
            const uint32_t noaer = pk_sub_sat(C.k_aer_nir, n) | C.force_noaer;                  // nir NOT <= 1000
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int j = wd * 2 + hf, sh = 16 * hf;
                const int bw = j >> 2, bk = j & 3;
                // ---- A4 quotient tests, sign-bit form (see the header comment)
                const int in1 = s16_of(n1, hf), id1 = s16_of(d1, hf), in2 = s16_of(n2, hf), id2 = s16_of(d2, hf);
                const double dn1 = (double)in1, dd1 = (double)id1, dn2 = (double)in2, dd2 = (double)id2;
                const double r0 = __builtin_fma(-P.qt[0], dd1, dn1), r1 = __builtin_fma(-P.qt[1], dd1, dn1),
                             r2 = __builtin_fma(-P.qt[2], dd1, dn1), r3 = __builtin_fma(-P.qt[3], dd2, dn2);
                // sign(h*d - r) = 1  <=>  r > h*d ;  sign(r - hneg*d) = 1  <=>  r < hneg*d ; exact zero -> +0
                const uint32_t g0 = hi32(__builtin_fma(P.qh[0], dd1, -r0)) >> 31, g1 = hi32(__builtin_fma(P.qh[1], dd1, -r1)) >> 31,
                               g2 = hi32(__builtin_fma(P.qh[2], dd1, -r2)) >> 31, l3 = hi32(__builtin_fma(-P.qh[3], dd2, r3)) >> 31;
                const uint32_t neg1 = (uint32_t)id1 >> 31, neg2 = (uint32_t)id2 >> 31;
                const uint32_t t1 = g0 ^ neg1, m_p1 = g1 ^ neg1, m_p2 = g2 ^ neg1, v_p1 = l3 ^ neg2;
                // ---- AWESH as int32: sign set <=> 4*awesh < awesh4_min  (T3 false)
                const int aw = C.awesh_init + 4 * s16_of(b, hf) + 10 * s16_of(g, hf) - 6 * s16_of(mn, hf) - s16_of(s2, hf);
                const uint32_t t3n = (uint32_t)aw >> 31;
                const uint32_t t2 = (t2s >> (15 + sh)) & 1u;
                const uint32_t t4 = m_p1 & v_p1 & ~(viol4 >> (15 + sh)) & 1u;
                const uint32_t t5 = m_p2 & ~(viol5 >> (15 + sh)) & 1u;
                const uint32_t fm = (vf[bw] >> (8 * bk)) & 0xffu;
                const uint32_t F = s_fm16[fm];
                const uint32_t band_ok = (bandvalid >> sh) & 1u;
                const uint32_t invalid = (band_ok ^ 1u) | ((F >> 8) & 1u);
                uint32_t ocean_nz = 1u, shad_nz = 1u, lbits = 0u;
                if (MASKS) {
                    ocean_nz = min((vo[bw] >> (8 * bk)) & 0xffu, 1u);
                    shad_nz = min((vs[bw] >> (8 * bk)) & 0xffu, 1u);
                    if (has_l) lbits = s_land8[(vl[bw] >> (8 * bk)) & 0xffu];
                }
                const uint32_t idx1 = t1 | t2 << 1 | t3n << 2 | t4 << 3 | t5 << 4 | invalid << 5 | (ocean_nz ^ 1u) << 6;
                const uint32_t word1 = s_lut1[idx1];
                const uint32_t code = (word1 >> 16) & 7u;
                const uint32_t remap = (F >> code) & ~(noaer >> (15 + sh)) & 1u;
                uint32_t idx2 = code | remap << 3 | ((F >> 5) & 7u) << 4;
                if (MASKS) {
                    const uint32_t shadrule = (shad_nz ^ 1u) & ~lbits & 1u;
                    const uint32_t lcpsw = (lbits >> 1) & ~(dark >> (15 + sh)) & 1u;
                    idx2 |= shadrule << 7 | lcpsw << 8 | ((lbits >> 2) & 1u) << 9;
                }
                const uint2 ch = s_chain[idx2];
                w1w[j] = word1; chx[j] = ch.x; chy[j] = ch.y;
                // ---- A3
                const uint32_t valid = (invalid ^ 1u) & ocean_nz & (in_range ? 1u : 0u);
                cnt += valid + ((valid & (F >> 9)) << 16);
            }
        }
}

// table words of 8 pixels -> plane dwords, in the order DIAG[4], WTR-1[2], then (lo, hi)
// pairs of WTR-1-AEROSOL, WTR-2, WTR, BWTR, CONF, CLOUD
struct GroupPlanes { uint32_t diag[4], w1[2], w1a[2], w2[2], w[2], bw[2], cf[2], cl[2]; };
__device__ __forceinline__ void lut_pack(const uint32_t (&w1w)[8], const uint32_t (&chx)[8], const uint32_t (&chy)[8],
                                         GroupPlanes& g) {
#pragma unroll
    for (int k = 0; k < 4; ++k) g.diag[k] = perm_b32(w1w[2 * k + 1], w1w[2 * k], 0x05040100u);
    g.w1[0] = perm_b32(perm_b32(w1w[3], w1w[2], 0x0c0c0703u), perm_b32(w1w[1], w1w[0], 0x0c0c0703u), 0x05040100u);
    g.w1[1] = perm_b32(perm_b32(w1w[7], w1w[6], 0x0c0c0703u), perm_b32(w1w[5], w1w[4], 0x0c0c0703u), 0x05040100u);
    uint32_t pa[4], pb[4], qa[4], qb[4];
    transpose4(chx, pa); transpose4(chx + 4, pb);
    transpose4(chy, qa); transpose4(chy + 4, qb);
    g.w1a[0] = pa[0]; g.w1a[1] = pb[0]; g.w2[0] = pa[1]; g.w2[1] = pb[1];
    g.w[0] = pa[2]; g.w[1] = pb[2]; g.bw[0] = pa[3]; g.bw[1] = pb[3];
    g.cf[0] = qa[0]; g.cf[1] = qb[0]; g.cl[0] = qa[1]; g.cl[1] = qb[1];
}

// ------------------------------------------------------------------------------
// Warp-specialised data movement + table-driven compute (DSWX_FUSED_VARIANT=4): phases A
// and C of dswx_classify_ws (LDS-DMA plane segments in, plane-run stores out) around
// lut_group / lut_pack.  LDS: 26 KiB images (32 KiB with masks) + 2.3 / 9.3 KiB of tables.
// ------------------------------------------------------------------------------
// ABLATE (diagnostic builds only, outputs meaningless): 1 = compute replaced by an xor fold,
// 2 = additionally no table loads, 3 = additionally no second barrier / partials
template <bool MASKS, int WPS, int ABLATE = 0>
__global__ __launch_bounds__(256, WPS) void dswx_classify_wslut(const KArgs a, const LutConsts C,
                                                               const Tables* __restrict__ tabs) {
    constexpr int N_CHAIN = MASKS ? 1024 : 128;
    __shared__ __attribute__((aligned(16))) uint8_t lds[WS_IN_MASKS + (MASKS ? 3 * WS_U8_BYTES : 0)];
    __shared__ uint32_t s_lut1[128];
    __shared__ uint16_t s_fm16[256];
    __shared__ uint8_t s_land8[MASKS ? 256 : 4];
    __shared__ uint2 s_chain[N_CHAIN];
    const DevParams& P = a.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile_base = (long long)blockIdx.y * a.n_pixels;
    const long long px0 = (long long)blockIdx.x * WS_PX;
    const long long n_vec = (a.n_pixels >> 3) << 3;
    const long long last16_i16 = (n_vec - 8) * 2, last16_u8 = n_vec >= 16 ? n_vec - 16 : 0;
    const bool has_l = MASKS && a.in.land, has_s = MASKS && a.in.shad, has_o = MASKS && a.in.ocean;

    // ---- phase A: LDS-DMA of the input planes (as dswx_classify_ws), tables by plain loads
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int piece = wave * 8 + q;
        if (piece < 24) {
            const int plane = piece >> 2, sub = piece & 3;
            long long byte = (px0 * 2) + sub * 1024 + lane * 16;
            byte = byte <= last16_i16 ? byte : last16_i16;
            __builtin_amdgcn_global_load_lds(
                (gptr_t)(reinterpret_cast<const uint8_t*>(a.in.band[plane]) + tile_base * 2 + byte),
                (lptr_t)(lds + plane * WS_BAND_BYTES + sub * 1024), 16, 0, 2);
        } else {
            const int u = (piece - 24) >> 1, sub = (piece - 24) & 1;
            const uint8_t* src = u == 0 ? a.in.fmask : (u == 1 ? a.in.land : (u == 2 ? a.in.shad : a.in.ocean));
            const bool present = u == 0 || (MASKS && ((u == 1 && has_l) || (u == 2 && has_s) || (u == 3 && has_o)));
            if (present) {
                long long byte = px0 + sub * 1024 + lane * 16;
                byte = byte <= last16_u8 ? byte : last16_u8;
                __builtin_amdgcn_global_load_lds((gptr_t)(src + tile_base + byte),
                                                 (lptr_t)(lds + WS_IN_FMASK + u * WS_U8_BYTES + sub * 1024), 16, 0, 2);
            }
        }
    }
    if (ABLATE < 2) {
    for (int i = threadIdx.x; i < 128; i += 256) s_lut1[i] = tabs->lut1[i];
    for (int i = threadIdx.x; i < 128; i += 256) reinterpret_cast<uint32_t*>(s_fm16)[i] = reinterpret_cast<const uint32_t*>(tabs->fm16)[i];
    if (MASKS) for (int i = threadIdx.x; i < 64; i += 256) reinterpret_cast<uint32_t*>(s_land8)[i] = reinterpret_cast<const uint32_t*>(tabs->land8)[i];
    for (int i = threadIdx.x; i < N_CHAIN; i += 256) s_chain[i] = tabs->chain[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- phase B
    const long long grp = (px0 >> 3) + threadIdx.x;
    const bool in_range = grp < (a.n_pixels >> 3);
    u32x4 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = *reinterpret_cast<const u32x4*>(lds + k * WS_BAND_BYTES + threadIdx.x * 16);
    const u32x2 vf = *reinterpret_cast<const u32x2*>(lds + WS_IN_FMASK + threadIdx.x * 8);
    u32x2 vl = {0u, 0u}, vs = {0x01010101u, 0x01010101u}, vo = {0x01010101u, 0x01010101u};
    uint32_t cnt = 0, t_ocean = 0;
    if (MASKS) {
        if (has_l) vl = *reinterpret_cast<const u32x2*>(lds + WS_IN_MASKS + threadIdx.x * 8);
        if (has_s) vs = *reinterpret_cast<const u32x2*>(lds + WS_IN_MASKS + WS_U8_BYTES + threadIdx.x * 8);
        if (has_o) {
            vo = *reinterpret_cast<const u32x2*>(lds + WS_IN_MASKS + 2 * WS_U8_BYTES + threadIdx.x * 8);
            t_ocean = __builtin_amdgcn_sad_u8(vo.x, 0u, __builtin_amdgcn_sad_u8(vo.y, 0u, 0u));
            t_ocean = in_range ? t_ocean : 0u;
        }
    }
    if (ABLATE < 3) __syncthreads();                     // the input images are dead from here on
    uint32_t w1w[8], chx[8], chy[8];
    if (ABLATE == 0) {
        lut_group<MASKS>(P, C, s_lut1, s_fm16, s_land8, s_chain, v, vf, vl, vs, vo, has_l, in_range, w1w, chx, chy, cnt);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t x = v[0][j >> 1] ^ v[1][j >> 1] ^ v[2][j >> 1] ^ v[3][j >> 1] ^ v[4][j >> 1] ^ v[5][j >> 1] ^ vf[j >> 2];
            w1w[j] = x; chx[j] = x + 1u; chy[j] = x + 2u;
        }
    }
    GroupPlanes gp;
    lut_pack(w1w, chx, chy, gp);
    *reinterpret_cast<u32x4*>(lds + threadIdx.x * 16) = u32x4{gp.diag[0], gp.diag[1], gp.diag[2], gp.diag[3]};
    uint8_t* su8 = lds + WS_OUT_U8 + threadIdx.x * 8;
    *reinterpret_cast<u32x2*>(su8 + 0 * WS_U8_BYTES) = u32x2{gp.w1[0], gp.w1[1]};
    if (a.out.wtr1_aerosol) *reinterpret_cast<u32x2*>(su8 + 1 * WS_U8_BYTES) = u32x2{gp.w1a[0], gp.w1a[1]};
    *reinterpret_cast<u32x2*>(su8 + 2 * WS_U8_BYTES) = u32x2{gp.w2[0], gp.w2[1]};
    *reinterpret_cast<u32x2*>(su8 + 3 * WS_U8_BYTES) = u32x2{gp.w[0], gp.w[1]};
    *reinterpret_cast<u32x2*>(su8 + 4 * WS_U8_BYTES) = u32x2{gp.bw[0], gp.bw[1]};
    *reinterpret_cast<u32x2*>(su8 + 5 * WS_U8_BYTES) = u32x2{gp.cf[0], gp.cf[1]};
    *reinterpret_cast<u32x2*>(su8 + 6 * WS_U8_BYTES) = u32x2{gp.cl[0], gp.cl[1]};
    if (a.partials && ABLATE < 3) {
        uint32_t c0 = cnt, c2 = t_ocean;
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) { c0 += __shfl_xor(c0, sh); c2 += __shfl_xor(c2, sh); }
        if (lane == 0) {
            const long long slot = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
            a.partials[slot] = make_uint2(c0, c2);
        }
    }
    __syncthreads();

    // ---- phase C (as dswx_classify_ws)
    const int diag_pieces = a.n_diag_pieces ? 4 : 0;
    const int n_pieces = diag_pieces + 2 * a.n_u8_out;
    const int per_wave = (n_pieces + 3) / 4;
    for (int q = 0; q < per_wave; ++q) {
        const int piece = wave * per_wave + q;
        if (piece >= n_pieces) break;
        if (piece < diag_pieces) {
            const long long p = px0 + piece * 512 + lane * 8;
            if (p + 8 <= n_vec)
                stg<u32x4, true>(a.out.diag + tile_base + p, *reinterpret_cast<const u32x4*>(lds + piece * 1024 + lane * 16));
        } else {
            const int u = (piece - diag_pieces) >> 1, sub = (piece - diag_pieces) & 1;
            const int region = a.u8_region[u];
            uint8_t* dst = a.u8_out[u] + tile_base;
            const long long p = px0 + sub * 1024 + lane * 16;
            const uint8_t* src = lds + WS_OUT_U8 + region * WS_U8_BYTES + sub * 1024 + lane * 16;
            if (p + 16 <= n_vec) stg<u32x4, true>(dst + p, *reinterpret_cast<const u32x4*>(src));
            else if (p + 8 <= n_vec) stg<u32x2, true>(dst + p, *reinterpret_cast<const u32x2*>(src));
        }
    }
}

// LUT_CHUNKS: 2048-px chunks per block (amortises the table load); WPS: launch bound
template <bool MASKS, int LUT_CHUNKS, int WPS>
__global__ __launch_bounds__(256, WPS) void dswx_classify_lut(const KArgs a, const LutConsts C,
                                                             const Tables* __restrict__ tabs) {
    constexpr int N_CHAIN = MASKS ? 1024 : 128;
    __shared__ uint32_t s_lut1[128];
    __shared__ uint16_t s_fm16[256];
    __shared__ uint8_t s_land8[MASKS ? 256 : 4];
    __shared__ uint2 s_chain[N_CHAIN];
    for (int i = threadIdx.x; i < 128; i += 256) s_lut1[i] = tabs->lut1[i];
    for (int i = threadIdx.x; i < 128; i += 256) reinterpret_cast<uint32_t*>(s_fm16)[i] = reinterpret_cast<const uint32_t*>(tabs->fm16)[i];
    if (MASKS) for (int i = threadIdx.x; i < 64; i += 256) reinterpret_cast<uint32_t*>(s_land8)[i] = reinterpret_cast<const uint32_t*>(tabs->land8)[i];
    for (int i = threadIdx.x; i < N_CHAIN; i += 256) s_chain[i] = tabs->chain[i];
    __syncthreads();

    const DevParams& P = a.P;
    const long long n_groups = a.n_pixels >> 3;
    const long long tile_base = (long long)blockIdx.y * a.n_pixels;
    const bool has_l = MASKS && a.in.land, has_s = MASKS && a.in.shad, has_o = MASKS && a.in.ocean;
    uint32_t cnt = 0, t_ocean = 0;       // cnt: valid in the low half, cloud-and-valid in the high half

    for (int c = 0; c < LUT_CHUNKS; ++c) {
        const long long grp = ((long long)blockIdx.x * LUT_CHUNKS + c) * 256 + threadIdx.x;
        if ((grp - threadIdx.x) >= n_groups) break;                       // block-uniform
        const bool in_range = grp < n_groups;
        const long long off = tile_base + (in_range ? grp : n_groups - 1) * 8;
        u32x4 v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = ldg<u32x4, true>(a.in.band[k] + off);
        const u32x2 vf = ldg<u32x2, true>(a.in.fmask + off);
        u32x2 vl = {0u, 0u}, vs = {0x01010101u, 0x01010101u}, vo = {0x01010101u, 0x01010101u};
        if (MASKS) {
            if (has_l) vl = ldg<u32x2, true>(a.in.land + off);
            if (has_s) vs = ldg<u32x2, true>(a.in.shad + off);
            if (has_o) {
                vo = ldg<u32x2, true>(a.in.ocean + off);
                const uint32_t so = __builtin_amdgcn_sad_u8(vo.x, 0u, __builtin_amdgcn_sad_u8(vo.y, 0u, 0u));
                t_ocean += in_range ? so : 0u;
            }
        }
        uint32_t w1w[8], chx[8], chy[8];      // per-pixel table words
        lut_group<MASKS>(P, C, s_lut1, s_fm16, s_land8, s_chain, v, vf, vl, vs, vo, has_l, in_range, w1w, chx, chy, cnt);
        if (in_range) {
            GroupPlanes gp;
            lut_pack(w1w, chx, chy, gp);
            if (a.out.diag) stg<u32x4, true>(a.out.diag + off, u32x4{gp.diag[0], gp.diag[1], gp.diag[2], gp.diag[3]});
            if (a.out.wtr1) stg<u32x2, true>(a.out.wtr1 + off, u32x2{gp.w1[0], gp.w1[1]});
            if (a.out.wtr1_aerosol) stg<u32x2, true>(a.out.wtr1_aerosol + off, u32x2{gp.w1a[0], gp.w1a[1]});
            if (a.out.wtr2) stg<u32x2, true>(a.out.wtr2 + off, u32x2{gp.w2[0], gp.w2[1]});
            if (a.out.wtr) stg<u32x2, true>(a.out.wtr + off, u32x2{gp.w[0], gp.w[1]});
            if (a.out.bwtr) stg<u32x2, true>(a.out.bwtr + off, u32x2{gp.bw[0], gp.bw[1]});
            if (a.out.conf) stg<u32x2, true>(a.out.conf + off, u32x2{gp.cf[0], gp.cf[1]});
            if (a.out.cloud) stg<u32x2, true>(a.out.cloud + off, u32x2{gp.cl[0], gp.cl[1]});
        }
    }
    if (a.partials) {
        uint32_t c0 = cnt, c2 = t_ocean;
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) { c0 += __shfl_xor(c0, sh); c2 += __shfl_xor(c2, sh); }
        if ((threadIdx.x & 63) == 0) {
            const long long slot = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
            a.partials[slot] = make_uint2(c0, c2);
        }
    }
}

// ------------------------------------------------------------------------------
// 'cover' mode, stage 2 (_add_snow_to_cloud_layer :2055-2078, then A11-A15).
//   snow  = dilate^10(Fmask bit 4)            restricted to  area = adjacent & (CLOUD == 0)
//   clear = dilate^7(~snow & (CLOUD == 0))    restricted to  area & (WTR-2 in 1..4)
//   snow &= ~clear
// with scipy.ndimage.binary_dilation semantics: 4-neighbour cross, synchronous
// iterations, cells outside the mask keep their value, outside the raster = False.
// k iterations reach k pixels, so a 64 x 64 output tile is exact from a 98 x 98
// window (halo 10 + 7) held in LDS; the two masked dilations ping-pong between
// two byte arrays there.  One block = 256 threads; grid = (tiles_x, tiles_y, n_tiles).
// ------------------------------------------------------------------------------
constexpr int CV_TILE = 64, CV_HALO = 17, CV_DIM = CV_TILE + 2 * CV_HALO, CV_CELLS = CV_DIM * CV_DIM;
enum : uint8_t { CV_SNOW = 1, CV_AREA = 2, CV_WATER = 4, CV_CLEAR0 = 8 };

__global__ __launch_bounds__(256) void dswx_cover_stage2(const KArgs a) {
    __shared__ uint8_t flags[CV_CELLS];      // static per-cell bits (CV_AREA, CV_WATER, CV_CLEAR0)
    __shared__ uint8_t cur[CV_CELLS];        // the mask being dilated
    __shared__ uint8_t nxt[CV_CELLS];
    const int H = a.height, W = a.width;
    const long long tile_base = (long long)blockIdx.z * a.n_pixels;
    const int y0 = blockIdx.y * CV_TILE - CV_HALO, x0 = blockIdx.x * CV_TILE - CV_HALO;
    for (int c = threadIdx.x; c < CV_CELLS; c += 256) {
        const int y = y0 + c / CV_DIM, x = x0 + c % CV_DIM;
        uint8_t f = 0, snow = 0;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const long long off = tile_base + (long long)y * W + x;
            const uint32_t fm = a.in.fmask[off], pc = a.cover_pc[off], w2 = a.cover_w2[off];
            snow = (fm & 16u) ? 1 : 0;
            const bool area = (fm & 4u) && pc == 0u;
            f = (area ? CV_AREA : 0) | ((w2 - 1u) <= 3u ? CV_WATER : 0) | (pc == 0u ? CV_CLEAR0 : 0);
        }
        flags[c] = f;
        cur[c] = snow;
    }
    __syncthreads();
    uint8_t* src = cur;
    uint8_t* dst = nxt;
    auto dilate = [&](uint8_t need) {
        for (int c = threadIdx.x; c < CV_CELLS; c += 256) {
            uint8_t v = src[c];
            if (!v && (flags[c] & need) == need) {
                const int yy = c / CV_DIM, xx = c % CV_DIM;
                v = (yy > 0 && src[c - CV_DIM]) || (yy < CV_DIM - 1 && src[c + CV_DIM]) ||
                    (xx > 0 && src[c - 1]) || (xx < CV_DIM - 1 && src[c + 1]);
            }
            dst[c] = v;
        }
        __syncthreads();
        uint8_t* t = src; src = dst; dst = t;
    };
    for (int it = 0; it < 10; ++it) dilate(CV_AREA);
    // src = dilated snow.  Keep it in `flags` (bit CV_SNOW) and start the second mask.
    for (int c = threadIdx.x; c < CV_CELLS; c += 256) {
        const uint8_t sn = src[c];
        const uint8_t f = flags[c];
        flags[c] = f | (sn ? CV_SNOW : 0);
        dst[c] = (!sn && (f & CV_CLEAR0)) ? 1 : 0;
    }
    __syncthreads();
    { uint8_t* t = src; src = dst; dst = t; }
    for (int it = 0; it < 7; ++it) dilate(CV_AREA | CV_WATER);
    // finish the interior 64 x 64
    for (int i = threadIdx.x; i < CV_TILE * CV_TILE; i += 256) {
        const int ly = i / CV_TILE, lx = i % CV_TILE;
        const int y = blockIdx.y * CV_TILE + ly, x = blockIdx.x * CV_TILE + lx;
        if (y >= H || x >= W) continue;
        const int c = (ly + CV_HALO) * CV_DIM + lx + CV_HALO;
        const bool snow = (flags[c] & CV_SNOW) && !src[c];
        const long long off = tile_base + (long long)y * W + x;
        PxOut o;
        finish_px(a.P, a.cover_w2[off], a.cover_pc[off], snow, o);
        if (a.out.wtr) a.out.wtr[off] = (uint8_t)o.wtr;
        if (a.out.bwtr) a.out.bwtr[off] = (uint8_t)o.bwtr;
        if (a.out.conf) a.out.conf[off] = (uint8_t)o.conf;
        if (a.out.cloud) a.out.cloud[off] = (uint8_t)o.cloud;
        if (a.out.browse) a.out.browse[off] = (uint8_t)o.browse;
    }
}

// Sums the fused kernel's per-wave partial counts of one tile (block = tile) and
// WRITES counters[tile]; the ragged-remainder kernel adds to them afterwards.
__global__ __launch_bounds__(256) void dswx_counters_finish(const uint2* __restrict__ partials,
                                                            unsigned long long* __restrict__ counters,
                                                            long long per_tile, int has_ocean,
                                                            long long vec_pixels) {
    __shared__ unsigned long long red[4][3];
    const uint2* p = partials + (long long)blockIdx.x * per_tile;
    unsigned long long v = 0, c = 0, o = 0;
    for (long long i = threadIdx.x; i < per_tile; i += 256) {
        const uint2 x = p[i];
        v += x.x & 0xffffu; c += x.x >> 16; o += x.y;
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
        v += __shfl_xor(v, sh); c += __shfl_xor(c, sh); o += __shfl_xor(o, sh);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = v; red[threadIdx.x >> 6][1] = c; red[threadIdx.x >> 6][2] = o; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* dst = counters + (long long)blockIdx.x * 3;
        dst[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        dst[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        dst[2] = has_ocean ? red[0][2] + red[1][2] + red[2][2] + red[3][2] : (unsigned long long)vec_pixels;
    }
}

// ------------------------------------------------------------------------------
// Generic kernel: one pixel per thread over pixels [px_begin, n_pixels) of every
// tile, no alignment requirement.  Runs the ragged remainder behind the vector
// kernel, and whole tiles when a batch of ragged tiles (n_pixels % 8 != 0,
// n_tiles > 1) or unaligned planes rule the vector kernel out.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dswx_classify_v1(const KArgs a) {
    __shared__ uint32_t lut32[64];
    __shared__ uint32_t red[4 * 3];
    if (threadIdx.x < 64) lut32[threadIdx.x] = a.P.aer_lut[threadIdx.x];
    __syncthreads();
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lut32);
    uint32_t c0 = 0, c1 = 0, c2 = 0;
    const long long px = a.px_begin + (long long)blockIdx.x * 256 + threadIdx.x;
    if (px < a.n_pixels) {
        const long long off = (long long)blockIdx.y * a.n_pixels + px;
        int land = -1, shad = 1, ocean = 1;
        if (a.in.land) land = a.in.land[off];
        if (a.in.shad) shad = a.in.shad[off];
        if (a.in.ocean) ocean = a.in.ocean[off];
        PxOut o;
        bool ok, cv;
        const int fm = a.in.fmask[off];
        classify_px(a.P, lut[fm], a.in.band[0][off], a.in.band[1][off], a.in.band[2][off], a.in.band[3][off],
                    a.in.band[4][off], a.in.band[5][off], fm, land, shad, ocean, o, ok, cv);
        c0 = ok ? 1u : 0u; c1 = cv ? 1u : 0u; c2 = (uint32_t)ocean;
        if (a.out.diag) a.out.diag[off] = (uint16_t)o.diag;
        if (a.out.wtr1) a.out.wtr1[off] = (uint8_t)o.wtr1;
        if (a.out.wtr1_aerosol) a.out.wtr1_aerosol[off] = (uint8_t)o.wtr1a;
        if (a.out.wtr2) a.out.wtr2[off] = (uint8_t)o.wtr2;
        if (a.out.wtr) a.out.wtr[off] = (uint8_t)o.wtr;
        if (a.out.bwtr) a.out.bwtr[off] = (uint8_t)o.bwtr;
        if (a.out.conf) a.out.conf[off] = (uint8_t)o.conf;
        if (a.out.cloud) a.out.cloud[off] = (uint8_t)o.cloud;
        if (a.out.browse) a.out.browse[off] = (uint8_t)o.browse;
        if (a.cover_w2) { a.cover_w2[off] = (uint8_t)o.w2_raw; a.cover_pc[off] = (uint8_t)o.pc; }
    }
    if (a.counters) reduce_counters(a.counters + (long long)blockIdx.y * 3, red, c0, c1, c2);
}

// ------------------------------------------------------------------------------
// Roofline probe: the fused kernel's plane traffic (six int16 planes + one u8 plane
// in, one u16 + six u8 planes out) with trivial math, in several access shapes, to
// measure the HBM rate each shape can reach at all.  Outputs are meaningless.
//   PPT   pixels per thread per iteration (8: 16-B int16 / 8-B u8 accesses;
//         16: 2x16-B int16 / 16-B u8 accesses)
//   NT    non-temporal loads and stores
// Each block walks `iters` consecutive chunks of 256*PPT pixels.
// ------------------------------------------------------------------------------
// MODE 0: read + write, 1: reads only, 2: writes only.  XCDMAP: block b works on
// chunk (b % 8) * ceil(nb / 8) + b / 8, i.e. every XCD walks its own contiguous
// eighth of the tile (blocks are dealt round-robin over the 8 XCDs).
template <int PPT, bool NT, int MODE, bool XCDMAP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void dswx_stream_probe_k(const KArgs a, int iters) {
    const long long n_groups = a.n_pixels / PPT;
    long long bx = blockIdx.x;
    if (XCDMAP) {
        const long long per = (gridDim.x + 7) / 8;
        bx = (bx & 7) * per + (bx >> 3);
    }
    for (int it = 0; it < iters; ++it) {
        const long long grp = (bx * iters + it) * BLOCK + threadIdx.x;
        if (grp >= n_groups) return;
        const long long off = (long long)blockIdx.y * a.n_pixels + grp * PPT;
        if (PPT == 8) {
            u32x4 x = {1u, 2u, 3u, (uint32_t)grp};
            u32x2 f = {5u, 6u};
            if (MODE != 2) {
                x = ldg<u32x4, NT>(a.in.band[0] + off);
#pragma unroll
                for (int k = 1; k < 6; ++k) x ^= ldg<u32x4, NT>(a.in.band[k] + off);
                f = ldg<u32x2, NT>(a.in.fmask + off);
            }
            u32x2 y; y.x = x.x ^ x.z ^ f.x; y.y = x.y ^ x.w ^ f.y;
            if (MODE == 1) {
                if (y.x == 0x12345678u && y.y == 0x9abcdef0u) stg<u32x2, NT>(a.out.wtr1 + off, y);
                continue;
            }
            stg<u32x4, NT>(a.out.diag + off, x);
            stg<u32x2, NT>(a.out.wtr1 + off, y);
            stg<u32x2, NT>(a.out.wtr2 + off, y + 1u);
            stg<u32x2, NT>(a.out.wtr + off, y + 2u);
            stg<u32x2, NT>(a.out.bwtr + off, y + 3u);
            stg<u32x2, NT>(a.out.conf + off, ~y);
            stg<u32x2, NT>(a.out.cloud + off, y + 5u);
        } else {
            u32x4 x0 = {1u, 2u, 3u, (uint32_t)grp}, x1 = x0, f = x0;
            if (MODE != 2) {
                x0 = ldg<u32x4, NT>(a.in.band[0] + off); x1 = ldg<u32x4, NT>(a.in.band[0] + off + 8);
#pragma unroll
                for (int k = 1; k < 6; ++k) {
                    x0 ^= ldg<u32x4, NT>(a.in.band[k] + off);
                    x1 ^= ldg<u32x4, NT>(a.in.band[k] + off + 8);
                }
                f = ldg<u32x4, NT>(a.in.fmask + off);
            }
            const u32x4 y = x0 ^ x1 ^ f;
            if (MODE == 1) {
                if (y.x == 0x12345678u && y.y == 0x9abcdef0u) stg<u32x4, NT>(a.out.wtr1 + off, y);
                continue;
            }
            stg<u32x4, NT>(a.out.diag + off, x0);
            stg<u32x4, NT>(a.out.diag + off + 8, x1);
            stg<u32x4, NT>(a.out.wtr1 + off, y);
            stg<u32x4, NT>(a.out.wtr2 + off, y + 1u);
            stg<u32x4, NT>(a.out.wtr + off, y + 2u);
            stg<u32x4, NT>(a.out.bwtr + off, y + 3u);
            stg<u32x4, NT>(a.out.conf + off, ~y);
            stg<u32x4, NT>(a.out.cloud + off, y + 5u);
        }
    }
}

// Staged probe: the fused kernel's data movement (register loads, LDS-transposed
// plane-run stores) with trivial math.  BLOCK threads x 8 px; each wave stores
// consecutive 1 KiB pieces.
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void dswx_staged_probe_k(const KArgs a) {
    constexpr int PX = BLOCK * 8, WAVES = BLOCK / 64;
    __shared__ __attribute__((aligned(16))) uint8_t stage[PX * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_groups = a.n_pixels >> 3;
    const long long grp = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const long long tile_base = (long long)blockIdx.y * a.n_pixels;
    const long long off = tile_base + (grp < n_groups ? grp : n_groups - 1) * 8;
    u32x4 x = ldg<u32x4, NT>(a.in.band[0] + off);
#pragma unroll
    for (int k = 1; k < 6; ++k) x ^= ldg<u32x4, NT>(a.in.band[k] + off);
    const u32x2 f = ldg<u32x2, NT>(a.in.fmask + off);
    u32x2 y; y.x = x.x ^ x.z ^ f.x; y.y = x.y ^ x.w ^ f.y;
    *reinterpret_cast<u32x4*>(stage + threadIdx.x * 16) = x;
#pragma unroll
    for (int k = 0; k < 6; ++k) *reinterpret_cast<u32x2*>(stage + PX * 2 + k * PX + threadIdx.x * 8) = y + (uint32_t)k;
    __syncthreads();
    uint8_t* const planes[7] = {reinterpret_cast<uint8_t*>(a.out.diag), a.out.wtr1, a.out.wtr2, a.out.wtr,
                                a.out.bwtr, a.out.conf, a.out.cloud};
    const long long px0 = (long long)blockIdx.x * PX, n_vec = n_groups * 8;
    constexpr int DIAG_PIECES = PX * 2 / 1024, U8_PIECES = PX / 1024, PIECES = DIAG_PIECES + 6 * U8_PIECES;
    constexpr int PER_WAVE = PIECES / WAVES;
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
        const int piece = wave * PER_WAVE + q;
        if (piece < DIAG_PIECES) {
            const long long p = px0 + piece * 512 + lane * 8;
            if (p + 8 <= n_vec) stg<u32x4, NT>(planes[0] + (tile_base + p) * 2, *reinterpret_cast<const u32x4*>(stage + piece * 1024 + lane * 16));
        } else {
            const int u = (piece - DIAG_PIECES) / U8_PIECES, sub = (piece - DIAG_PIECES) % U8_PIECES;
            const long long p = px0 + sub * 1024 + lane * 16;
            if (p + 16 <= n_vec) stg<u32x4, NT>(planes[1 + u] + tile_base + p, *reinterpret_cast<const u32x4*>(stage + PX * 2 + u * PX + sub * 1024 + lane * 16));
        }
    }
}

// Stream-count calibration: the same 14 planes and bytes, but every block streams
// 4 KiB of ONE plane (blockIdx.x % 14 selects it): 7 read-only streams and 7
// write-only streams that never meet inside a block.
template <bool NT>
__global__ __launch_bounds__(256) void dswx_plane_per_block_k(const KArgs a, long long total_px) {
    const int plane = blockIdx.x % 14;
    const long long chunk = blockIdx.x / 14;                 // 4 KiB chunk index within the plane
    const long long byte = chunk * 4096 + threadIdx.x * 16;
    if (plane < 7) {
        const uint8_t* src = plane < 6 ? reinterpret_cast<const uint8_t*>(a.in.band[plane]) : a.in.fmask;
        const long long bytes = plane < 6 ? total_px * 2 : total_px;
        // int16 planes are twice as long: walk two chunks
        u32x4 x = {0u, 0u, 0u, 0u};
        if (byte < bytes) x = ldg<u32x4, NT>(src + byte);
        if (plane < 6 && byte + bytes / 2 < bytes && byte < bytes / 2) x ^= ldg<u32x4, NT>(src + bytes / 2 + byte);
        if (x.x == 0x9E3779B9u && x.y == 0x7F4A7C15u) a.out.wtr1[0] = 1;
    } else {
        uint8_t* const planes[7] = {reinterpret_cast<uint8_t*>(a.out.diag), a.out.wtr1, a.out.wtr2, a.out.wtr,
                                    a.out.bwtr, a.out.conf, a.out.cloud};
        const int z = plane - 7;
        const long long bytes = z == 0 ? total_px * 2 : total_px;
        const u32x4 val = {threadIdx.x, blockIdx.x, 3u, 4u};
        if (byte < bytes) stg<u32x4, NT>(planes[z] + byte, val);
        if (z == 0 && byte + bytes / 2 < bytes && byte < bytes / 2) stg<u32x4, NT>(planes[0] + bytes / 2 + byte, val);
    }
}

// Role-split calibration, same planes and bytes as the fused kernel, 8 px per lane.
// SPLIT 0: even blocks read all 7 input planes (two chunks each), odd blocks write
// all 7 output planes (two chunks each).  SPLIT 1: inside every block waves 0-1
// only read (two chunks), waves 2-3 only write (two chunks).
template <int SPLIT, bool NT>
__global__ __launch_bounds__(256) void dswx_role_split_k(const KArgs a) {
    const long long n_groups = a.n_pixels >> 3;
    const long long tile_base = (long long)blockIdx.y * a.n_pixels;
    if (SPLIT == 2) {
        // pairs of blocks cover 4096 px: the even one reads all 7 planes (2 groups per
        // thread), the odd one writes all 7 planes as 1 KiB pieces, 8 consecutive
        // pieces per wave (the LDS-transposed store shape, without the LDS)
        // roles alternate every 8 blocks so that every XCD (blocks are dealt round-robin
        // over the 8 XCDs) hosts readers and writers alike
        const long long pair = (long long)(blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
        if (((blockIdx.x >> 3) & 1) == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const long long grp = pair * 512 + h * 256 + threadIdx.x;
                if (grp >= n_groups) continue;
                const long long off = tile_base + grp * 8;
                u32x4 x = ldg<u32x4, NT>(a.in.band[0] + off);
#pragma unroll
                for (int k = 1; k < 6; ++k) x ^= ldg<u32x4, NT>(a.in.band[k] + off);
                const u32x2 f = ldg<u32x2, NT>(a.in.fmask + off);
                if ((x.x ^ f.x) == 0x12345678u && (x.y ^ f.y) == 0x9abcdef0u) a.out.wtr1[0] = 1;
            }
        } else {
            uint8_t* const planes[7] = {reinterpret_cast<uint8_t*>(a.out.diag), a.out.wtr1, a.out.wtr2, a.out.wtr,
                                        a.out.bwtr, a.out.conf, a.out.cloud};
            const long long px0 = pair * 4096;
            if (px0 + 4096 > a.n_pixels) return;
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            const u32x4 val = {threadIdx.x, blockIdx.x, 3u, 4u};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int piece = wave * 8 + q;
                int plane, sub;
                if (piece < 8) { plane = 0; sub = piece; } else { plane = 1 + (piece - 8) / 4; sub = (piece - 8) % 4; }
                const long long byte0 = plane == 0 ? (tile_base + px0) * 2 : tile_base + px0;
                stg<u32x4, NT>(planes[plane] + byte0 + sub * 1024 + lane * 16, val);
            }
        }
        return;
    }
    bool reader;
    long long g0, g1;
    if (SPLIT == 0) {
        reader = ((blockIdx.x >> 3) & 1) == 0;                       // XCD-balanced roles
        const long long pair = (long long)(blockIdx.x >> 4) * 8 + (blockIdx.x & 7);   // groups [pair*512, +512)
        g0 = pair * 512 + threadIdx.x; g1 = g0 + 256;
    } else {
        reader = threadIdx.x < 128;
        const long long t = threadIdx.x & 127;
        g0 = (long long)blockIdx.x * 256 + t; g1 = g0 + 128;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long grp = h ? g1 : g0;
        if (grp >= n_groups) continue;
        const long long off = tile_base + grp * 8;
        if (reader) {
            u32x4 x = ldg<u32x4, NT>(a.in.band[0] + off);
#pragma unroll
            for (int k = 1; k < 6; ++k) x ^= ldg<u32x4, NT>(a.in.band[k] + off);
            const u32x2 f = ldg<u32x2, NT>(a.in.fmask + off);
            if ((x.x ^ f.x) == 0x12345678u && (x.y ^ f.y) == 0x9abcdef0u) a.out.wtr1[0] = 1;
        } else {
            const u32x4 x = {threadIdx.x, blockIdx.x, 3u, (uint32_t)grp};
            const u32x2 y = {x.x, x.w};
            stg<u32x4, NT>(a.out.diag + off, x);
            stg<u32x2, NT>(a.out.wtr1 + off, y);
            stg<u32x2, NT>(a.out.wtr2 + off, y + 1u);
            stg<u32x2, NT>(a.out.wtr + off, y + 2u);
            stg<u32x2, NT>(a.out.bwtr + off, y + 3u);
            stg<u32x2, NT>(a.out.conf + off, ~y);
            stg<u32x2, NT>(a.out.cloud + off, y + 5u);
        }
    }
}

// Layout calibration: the fused kernel's thread mapping and bytes, but the 14 planes
// interleaved in chunks of CH pixels inside one arena: chunk c holds
// [6 x int16 | fmask | diag u16 | 6 x u8] for pixels [c*CH, (c+1)*CH), so the 14
// accesses of a block fall within one 21*CH-byte span instead of 14 distant planes.
template <int CH, bool NT>
__global__ __launch_bounds__(256) void dswx_chunked_layout_probe_k(uint8_t* __restrict__ arena, long long total_px) {
    const long long grp = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long px = grp * 8;
    if (px >= total_px) return;
    const long long c = px / CH, r = px % CH;
    uint8_t* base = arena + c * (21LL * CH);
    u32x4 x = ldg<u32x4, NT>(base + r * 2);
#pragma unroll
    for (int k = 1; k < 6; ++k) x ^= ldg<u32x4, NT>(base + (long long)k * 2 * CH + r * 2);
    const u32x2 f = ldg<u32x2, NT>(base + 12LL * CH + r);
    u32x2 y; y.x = x.x ^ x.z ^ f.x; y.y = x.y ^ x.w ^ f.y;
    stg<u32x4, NT>(base + 13LL * CH + r * 2, x);
#pragma unroll
    for (int k = 0; k < 6; ++k) stg<u32x2, NT>(base + (15LL + k) * CH + r, y + (uint32_t)k);
}

// Plane-specialised waves: block = 7 waves over a 4096-px chunk; wave k reads only input
// plane k (8 KiB of an int16 plane, 4 KiB of Fmask) and then writes only output plane k
// (8 KiB of DIAG, 4 KiB of a u8 layer).  Same bytes as the fused kernel; this is what a
// warp-specialised loader / storer design would present to the memory system.
template <bool NT>
__global__ __launch_bounds__(448) void dswx_plane_per_wave_k(const KArgs a, long long total_px) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long px0 = (long long)blockIdx.x * 4096;
    if (px0 + 4096 > total_px) return;
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (wave < 6) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(a.in.band[wave]) + px0 * 2;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc ^= ldg<u32x4, NT>(src + q * 1024 + lane * 16);
    } else {
        const uint8_t* src = a.in.fmask + px0;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc ^= ldg<u32x4, NT>(src + q * 1024 + lane * 16);
    }
    uint8_t* const planes[7] = {reinterpret_cast<uint8_t*>(a.out.diag), a.out.wtr1, a.out.wtr2, a.out.wtr,
                                a.out.bwtr, a.out.conf, a.out.cloud};
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) stg<u32x4, NT>(planes[0] + px0 * 2 + q * 1024 + lane * 16, acc + (uint32_t)q);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) stg<u32x4, NT>(planes[wave] + px0 + q * 1024 + lane * 16, acc + (uint32_t)q);
    }
}

// Warp-specialised data movement: block = 256 threads over a 2048-px chunk.
//  phase A  wave w pulls planes {w, w+4} of the chunk into LDS with LDS-DMA
//           (global_load_lds, 1 KiB per wave-instruction, no VGPR staging): each wave
//           reads 4 KiB (2 KiB for Fmask) of ONE plane contiguously;
//  phase B  every thread folds its 8 pixels out of the seven LDS images and parks
//           results in the output staging regions;
//  phase C  wave w writes whole plane runs (as the LDS-staged kernel does).
// LDS: 26 KiB in + 18 KiB out = 44 KiB per block (3 blocks per CU).
template <bool NT>
__global__ __launch_bounds__(256) void dswx_ws_probe_k(const KArgs a) {
    constexpr int PX = 2048;
    __shared__ __attribute__((aligned(16))) uint8_t lds_in[6 * PX * 2 + PX];
    __shared__ __attribute__((aligned(16))) uint8_t lds_out[PX * 2 + 6 * PX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile_base = (long long)blockIdx.y * a.n_pixels;
    const long long px0 = (long long)blockIdx.x * PX;
    if (px0 + PX > a.n_pixels) return;       // probe only: whole chunks
    // phase A
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int plane = wave + 4 * h;
        if (plane < 6) {
            const uint8_t* src = reinterpret_cast<const uint8_t*>(a.in.band[plane]) + (tile_base + px0) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + q * 1024 + lane * 16),
                                                 (lptr_t)(lds_in + plane * (PX * 2) + q * 1024), 16, 0, NT ? 2 : 0);
        } else if (plane == 6) {
            const uint8_t* src = a.in.fmask + tile_base + px0;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + q * 1024 + lane * 16),
                                                 (lptr_t)(lds_in + 6 * (PX * 2) + q * 1024), 16, 0, NT ? 2 : 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // phase B
    u32x4 x = *reinterpret_cast<const u32x4*>(lds_in + threadIdx.x * 16);
#pragma unroll
    for (int k = 1; k < 6; ++k) x ^= *reinterpret_cast<const u32x4*>(lds_in + k * (PX * 2) + threadIdx.x * 16);
    const u32x2 f = *reinterpret_cast<const u32x2*>(lds_in + 6 * (PX * 2) + threadIdx.x * 8);
    u32x2 y; y.x = x.x ^ x.z ^ f.x; y.y = x.y ^ x.w ^ f.y;
    *reinterpret_cast<u32x4*>(lds_out + threadIdx.x * 16) = x;
#pragma unroll
    for (int k = 0; k < 6; ++k) *reinterpret_cast<u32x2*>(lds_out + PX * 2 + k * PX + threadIdx.x * 8) = y + (uint32_t)k;
    __syncthreads();
    // phase C: 16 pieces of 1 KiB (diag 4, six u8 planes 2 each), 4 consecutive per wave
    uint8_t* const planes[7] = {reinterpret_cast<uint8_t*>(a.out.diag), a.out.wtr1, a.out.wtr2, a.out.wtr,
                                a.out.bwtr, a.out.conf, a.out.cloud};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q;
        if (piece < 4) {
            stg<u32x4, NT>(planes[0] + (tile_base + px0) * 2 + piece * 1024 + lane * 16,
                           *reinterpret_cast<const u32x4*>(lds_out + piece * 1024 + lane * 16));
        } else {
            const int u = (piece - 4) >> 1, sub = (piece - 4) & 1;
            stg<u32x4, NT>(planes[1 + u] + tile_base + px0 + sub * 1024 + lane * 16,
                           *reinterpret_cast<const u32x4*>(lds_out + PX * 2 + u * PX + sub * 1024 + lane * 16));
        }
    }
}

// Calibration: a flat two-stream copy moving the same 13 B in / 8 B out per pixel
// (reads `n16_in` 16-byte words from src, writes `n16_out` to dst).
template <bool NT>
__global__ __launch_bounds__(256) void dswx_flat_copy_k(const u32x4* __restrict__ src, u32x4* __restrict__ dst,
                                                        long long n16_in, long long n16_out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    u32x4 x = {0u, 0u, 0u, 0u};
    if (i < n16_in) x = ldg<u32x4, NT>(src + i);
    // every thread reads one word; the first n16_out threads also write one
    if (i < n16_out) stg<u32x4, NT>(dst + i, x);
    else if (x.x == 0x9E3779B9u && x.y == 0x7F4A7C15u) dst[0] = x;   // keep the load alive
}

// Write-path calibration (outputs meaningless).  WMODE 0: one flat stream of
// 16-byte stores; 1: seven planes, each BLOCK writes 4 KiB of ONE plane
// (blockIdx.z = plane); 2: seven planes, each WAVE of a block writes 1 KiB pieces
// of its own planes (the store shape an LDS-transposed epilogue would have).
template <int WMODE, bool NT>
__global__ __launch_bounds__(256) void dswx_write_probe_k(const KArgs a, long long total_px) {
    const u32x4 val = {threadIdx.x, blockIdx.x, 3u, 4u};
    if (WMODE == 0) {
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // 16-byte words
        if (i < total_px * 8 / 16) stg<u32x4, NT>(reinterpret_cast<u32x4*>(a.out.diag) + i, val);
    } else if (WMODE == 1) {
        // plane z: 0 = diag (2 B/px, two blocks' worth), 1..6 = u8 planes
        uint8_t* const planes[7] = {reinterpret_cast<uint8_t*>(a.out.diag), a.out.wtr1, a.out.wtr2, a.out.wtr,
                                    a.out.bwtr, a.out.conf, a.out.cloud};
        const int z = blockIdx.z;
        const long long bytes = z == 0 ? total_px * 2 : total_px;
        const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
        if (i < bytes) stg<u32x4, NT>(planes[z] + i, val);
        if (z == 0 && i + bytes / 2 < bytes && i < bytes / 2) stg<u32x4, NT>(planes[0] + bytes / 2 + i, val);
    } else {
        // block covers 4096 px: per u8 plane 4 KiB = 4 wave-stores of 1 KiB, diag 8 KiB = 8.
        // 32 wave-stores in all, 8 per wave: wave w writes diag quarter w (2) + planes
        // {w, w+4 (if < 6)} hmm -> keep it simple: wave w writes pieces p = w, w+4, ... of the
        // 32-piece list [diag x8, wtr1 x4, wtr2 x4, wtr x4, bwtr x4, conf x4, cloud x4]
        uint8_t* const planes[7] = {reinterpret_cast<uint8_t*>(a.out.diag), a.out.wtr1, a.out.wtr2, a.out.wtr,
                                    a.out.bwtr, a.out.conf, a.out.cloud};
        const long long px0 = (long long)blockIdx.x * 4096;
        if (px0 >= total_px) return;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int piece = wave * 8 + q;             // consecutive pieces: a wave stays in one or two planes
            int plane, sub;
            if (piece < 8) { plane = 0; sub = piece; } else { plane = 1 + (piece - 8) / 4; sub = (piece - 8) % 4; }
            const long long byte0 = plane == 0 ? px0 * 2 : px0;
            stg<u32x4, NT>(planes[plane] + byte0 + sub * 1024 + lane * 16, val);
        }
    }
}

// ------------------------------------------------------------------------------
// generate_interpreted_layer (:1687-1707) on its own: DIAG in decimal (any integer,
// as the reference's unit test feeds it) -> WTR-1 class; 32 and anything outside
// the table -> 255.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dswx_interpret_v1(const long long* __restrict__ diag,
                                                         uint8_t* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long d = diag[i];
    uint32_t cls = 255u;
    if (d >= 0 && d < 32) {
        const uint32_t k = (uint32_t)d;
        cls = ((CLS_B0 >> k) & 1u) | (((CLS_B1 >> k) & 1u) << 1) | (((CLS_B2 >> k) & 1u) << 2);
    }
    out[i] = (uint8_t)cls;
}

// ------------------------------------------------------------------------------
// Terrain shadow layer (row f1): _compute_opera_shadow_layer :4215-4283 followed by
// the margin crop of :4320 / :5170.  One thread per OUTPUT pixel.
//
// Arithmetic types follow what numpy >= 2 (NEP 50) gives the reference expressions on a
// float32 DEM: np.gradient, the division by the pixel spacing, the squares, their sum,
// `+ 1` and the sqrt stay float32; the products with the float64 sun-vector scalars,
// the quotient, arccos / arctan / degrees and the comparisons are float64.  (Under the
// numpy 1.23.5 the reference pins, value-based casting keeps those float32 as well:
// borderline pixels can differ between the two -- SURVEY.md §7.)  Built without
// fp contraction; hipcc's float32 division and sqrt are correctly rounded.
// ------------------------------------------------------------------------------
struct ShadowArgs {
    const float* dem;        // [H][W], with margin
    uint8_t* shadow;         // [H - 2*margin][W - 2*margin]; 1 = not shadow, 0 = shadow
    long long height, width, margin;
    float spacing_x, neg_abs_spacing_y;
    double sun[3];           // target-to-sun unit vector (x, y, z)
    double sin_az, cos_az;
    double min_slope_angle, max_sun_local_inc_angle;
};

__global__ __launch_bounds__(256) void dswx_shadow_v1(const ShadowArgs a) {
    const long long ow = a.width - 2 * a.margin, oh = a.height - 2 * a.margin;
    const long long ox = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    const long long oy = (long long)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= ow || oy >= oh) return;
    const long long x = ox + a.margin, y = oy + a.margin, W = a.width, H = a.height;
    const float* d = a.dem + (long long)blockIdx.z * H * W;
    // np.gradient, edge_order 1, unit spacing: central differences inside,
    // one-sided at the borders
    float gx, gy;
    if (x == 0) gx = d[y * W + 1] - d[y * W];
    else if (x == W - 1) gx = d[y * W + x] - d[y * W + x - 1];
    else gx = (d[y * W + x + 1] - d[y * W + x - 1]) / 2.0f;
    if (y == 0) gy = d[W + x] - d[x];
    else if (y == H - 1) gy = d[y * W + x] - d[(y - 1) * W + x];
    else gy = (d[(y + 1) * W + x] - d[(y - 1) * W + x]) / 2.0f;
    const float n0 = -gx / a.spacing_x;
    const float n1 = -gy / a.neg_abs_spacing_y;
    const float norm = sqrtf(n0 * n0 + n1 * n1 + 1.0f);
    const double dot = (double)n0 * a.sun[0] + (double)n1 * a.sun[1] + a.sun[2];
    const double RAD2DEG = 180.0 / 3.141592653589793238462643383279502884;
    const double inc_deg = acos(dot / (double)norm) * RAD2DEG;
    const double slope_deg = atan((double)n0 * a.sin_az + (double)n1 * a.cos_az) * RAD2DEG;
    const bool backslope = slope_deg <= a.min_slope_angle;
    const bool low_inc = inc_deg <= a.max_sun_local_inc_angle;
    a.shadow[(long long)blockIdx.z * oh * ow + oy * ow + ox] = (low_inc | !backslope) ? 1 : 0;
}

// ------------------------------------------------------------------------------
// LAND layer (row f3): the per-pixel part of create_landcover_mask :994-1115.
// One thread per HLS pixel: 3x3 WorldCover block -> three counts -> class hierarchy.
// ------------------------------------------------------------------------------
struct LandArgs {
    const uint8_t* wc3;      // [3H][3W]
    const uint8_t* cgls;     // [H][W]
    uint8_t* land;           // [H][W]
    long long height, width;
    uint32_t forest_bits[8]; // 256-bit set of CGLS forest classes
    int thr_tree, thr_low, thr_high, thr_water;
    int low_class, high_class;   // year_offset, 100 + year_offset (as uint8)
};

__global__ __launch_bounds__(256) void dswx_landcover_v1(const LandArgs a) {
    const long long x = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    const long long y = (long long)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.width || y >= a.height) return;
    int water = 0, urban = 0, tree = 0;
    const long long W3 = 3 * a.width;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint8_t* row = a.wc3 + (3 * y + i) * W3 + 3 * x;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int v = row[j];
            water += (v == 80) | (v == 90) | (v == 95);
            urban += v == 50;
            tree += v == 10;
        }
    }
    const int c = a.cgls[y * a.width + x];
    if (!((a.forest_bits[c >> 5] >> (c & 31)) & 1u)) tree = 0;
    int v = 255;
    if (tree >= a.thr_tree) v = 201;
    if (urban >= a.thr_low) v = a.low_class;
    if (urban >= a.thr_high) v = a.high_class;
    if (water >= a.thr_water) v = 200;
    a.land[y * a.width + x] = (uint8_t)v;
}

// ------------------------------------------------------------------------------
// Debug planes: float64 MNDWI / NDVI / AWESH exactly as :1872-1887 (true IEEE
// division; int16 wrap-around sums).  Not on the timed path.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dswx_indices_v1(const KArgs a, long long total) {
    const long long off = (long long)blockIdx.x * 256 + threadIdx.x;
    if (off >= total) return;
    int b = a.in.band[0][off], g = a.in.band[1][off], r = a.in.band[2][off], n = a.in.band[3][off],
        s1 = a.in.band[4][off], s2 = a.in.band[5][off];
    const int cm = a.P.clip_min;
    b = max(b, cm); g = max(g, cm); r = max(r, cm);
    n = max(n, cm); s1 = max(s1, cm); s2 = max(s2, cm);
    const int d1 = (short)(g + s1), n1 = (short)(g - s1), mbsrn = (short)(n + s1);
    const int n2 = (short)(n - r), d2 = (short)(n + r);
    if (a.out.mndwi) a.out.mndwi[off] = (double)n1 / (double)d1;
    if (a.out.ndvi) a.out.ndvi[off] = (double)n2 / (double)d2;
    if (a.out.awesh) a.out.awesh[off] = 0.25 * (double)(4 * b + 10 * g - 6 * mbsrn - s2);
}

// ------------------------------------------------------------------------------
// Synthetic tiles (same integer recipe as proteus_amd/synth.py)
// ------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ long long fieldu(unsigned long long h, int shift, int bits) {
    return (long long)((h >> shift) & ((1ull << bits) - 1ull));
}

__global__ __launch_bounds__(256) void dswx_synth_v1(dswx_planes_in_t in, unsigned long long seed,
                                                      long long tile0, long long n_pixels, int width) {
    const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
    if (px >= n_pixels) return;
    const long long t = blockIdx.y;
    const unsigned long long tile = (unsigned long long)(tile0 + t);
    const long long off = t * n_pixels + px;
    const unsigned long long K0 = 0x9E3779B97F4A7C15ull, K1 = 0xD1B54A32D192ED03ull;
    const unsigned long long h0 = mix64(seed * K0 + tile * K1 + (unsigned long long)px);
    const unsigned long long h1 = mix64(h0 + K0);
    const unsigned long long h2 = mix64(h1 + K0);
    const int cuts[5] = {14418, 26214, 42598, 55705, 64225};
    const int mean[5][6] = {{350, 450, 350, 250, 150, 100},
                            {500, 700, 600, 1300, 800, 500},
                            {300, 600, 400, 3500, 1800, 900},
                            {900, 1200, 1500, 2200, 2800, 2300},
                            {6000, 6200, 6400, 6600, 3000, 2500}};
    const int amps[5] = {300, 600, 600, 800, 2500};
    const int draw = (int)fieldu(h0, 0, 16);
    int stype = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) stype += draw >= cuts[k] ? 1 : 0;
    const bool is_fill = stype == 5;
    const int st = stype < 4 ? stype : 4;
    const long long amp = amps[st];
    const bool clip_evt = fieldu(h0, 16, 7) == 0;
    const int clip_band = (int)((fieldu(h0, 23, 3) * 6) >> 3);
    const long long clip_val = -fieldu(h0, 26, 8);
    const bool wrap_evt = fieldu(h0, 34, 10) == 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const long long noise = fieldu(h1, 10 * b, 10);
        long long v = mean[st][b] + ((noise * 2 * amp) >> 10) - amp;
        if (wrap_evt && (b == 1 || b == 4)) v += 19000;
        if (clip_evt && clip_band == b) v = clip_val;
        if (is_fill) v = -9999;
        const_cast<int16_t*>(in.band[b])[off] = (int16_t)v;
    }
    const long long aerosol = fieldu(h2, 0, 2);
    const long long water = fieldu(h2, 2, 5) < 10, snow = fieldu(h2, 7, 5) < 2,
                    shadow = fieldu(h2, 12, 5) < 3, adjacent = fieldu(h2, 17, 5) < 3,
                    cloud = fieldu(h2, 22, 5) < 4, cirrus = fieldu(h2, 27, 5) < 1;
    long long fm = (aerosol << 6) | (water << 5) | (snow << 4) | (shadow << 3) | (adjacent << 2) |
                   (cloud << 1) | cirrus;
    if (is_fill) fm = 255;
    const_cast<uint8_t*>(in.fmask)[off] = (uint8_t)fm;
    if (in.land) {
        const int classes[8] = {200, 201, 21, 121, 50, 150, 99, 100};
        const int cls = classes[fieldu(h2, 40, 3)];
        const_cast<uint8_t*>(in.land)[off] = (uint8_t)(fieldu(h2, 32, 8) < 179 ? 255 : cls);
    }
    if (in.shad) const_cast<uint8_t*>(in.shad)[off] = (uint8_t)(fieldu(h2, 43, 5) >= 3 ? 1 : 0);
    if (in.ocean) {
        const unsigned long long row_band = (unsigned long long)(px / width) >> 5;
        const unsigned long long hb = mix64(seed * K1 + tile * K0 + row_band + 0x5851F42D4C957F2Dull);
        const_cast<uint8_t*>(in.ocean)[off] = (uint8_t)(fieldu(hb, 0, 8) >= 13 ? 1 : 0);
    }
}

// ==============================================================================
// host side
// ==============================================================================
struct dswx_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    // grow-only staging for dswx_classify_host
    void* stage = nullptr;
    size_t stage_bytes = 0;
    // grow-only workspace for the vector kernel's per-wave counter partials
    void* partials = nullptr;
    size_t partials_bytes = 0;
    // device copy of the lookup tables of the table-driven kernel (rebuilt per call)
    void* tables = nullptr;
    // grow-only scratch of 'cover' mode: uncollapsed WTR-2 + pre-snow CLOUD planes
    void* cover = nullptr;
    size_t cover_bytes = 0;
    std::string last_kernel;
    int fused_variant = 0;   // env DSWX_FUSED_VARIANT -- 0: direct stores (default); 1: LDS-staged
                             // stores; 2: warp-specialised (LDS-DMA in, plane-run stores out);
                             // 3: table-driven (packed int16 + LDS tables + v_perm packing);
                             // 4: warp-specialised data movement + table-driven compute
    int tune_ablate = 0;     // diagnostic ablation level of variant 4 (env DSWX_TUNE_ABLATE; outputs invalid)
    int tune_chunks = 1;     // table-driven kernel: chunks per block (env DSWX_TUNE_CHUNKS: 1, 4)
    int tune_lut_wps = 5;    // table-driven kernel: launch bound (env DSWX_TUNE_LUT_WPS: 4, 5, 6)
    int tune_wps = 6;        // launch-bound variant of the plain kernel (env DSWX_TUNE_WPS: 4, 6, 8)
};

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return fail(DSWX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// smallest k with (x > t <=> x >= k) for every integer x in the int16-sum range
static int32_t int_gt_min(double t) {
    if (std::isnan(t)) return 1 << 30;
    double k = std::floor(t) + 1.0;
    if (k > 1e9) return 1 << 30;
    if (k < -1e9) return -(1 << 30);
    return (int32_t)k;
}
// largest k with (x < t <=> x <= k)
static int32_t int_lt_max(double t) {
    if (std::isnan(t)) return -(1 << 30);
    double k = std::ceil(t) - 1.0;
    if (k > 1e9) return 1 << 30;
    if (k < -1e9) return -(1 << 30);
    return (int32_t)k;
}
// largest k with (x <= t <=> x <= k)
static int32_t int_le_max(double t) {
    if (std::isnan(t)) return -(1 << 30);
    double k = std::floor(t);
    if (k > 1e9) return 1 << 30;
    if (k < -1e9) return -(1 << 30);
    return (int32_t)k;
}

static int make_dev_params(const dswx_params_t* p, DevParams* d) {
    const double thr[12] = {p->wigt, p->awgt, p->pswt_1_mndwi, p->pswt_1_nir, p->pswt_1_swir1,
                            p->pswt_1_ndvi, p->pswt_2_mndwi, p->pswt_2_blue, p->pswt_2_nir,
                            p->pswt_2_swir1, p->pswt_2_swir2, p->lcmask_nir};
    for (double t : thr)
        if (!std::isfinite(t) || std::fabs(t) > 1e100 || (t != 0.0 && std::fabs(t) < 1e-290))
            return fail(DSWX_ERR_ARG, "HLS thresholds must be finite, |t| <= 1e100, and 0 or |t| >= 1e-290");
    if (!std::isfinite(p->aerosol_max_nir)) return fail(DSWX_ERR_ARG, "aerosol_max_nir must be finite");
    if (p->mask_adjacent_to_cloud_mode < 0 || p->mask_adjacent_to_cloud_mode > 2)
        return fail(DSWX_ERR_UNSUPPORTED, "ERROR mask adjacent to cloud/cloud-shadow mode: %d",
                    p->mask_adjacent_to_cloud_mode);
    std::memset(d, 0, sizeof *d);
    const double inf = std::numeric_limits<double>::infinity();
    // Half gap to the neighbouring double.  At t == 0 the true half gap (2^-1075)
    // is not representable; any h with 0 < h*|d| < 1 <= |n| separates the same
    // quotients, so 2^-100 stands in for it.
    const double h_at_zero = std::ldexp(1.0, -100);
    const double gt_thr[3] = {p->wigt, p->pswt_1_mndwi, p->pswt_2_mndwi};
    for (int i = 0; i < 3; ++i) {
        d->qt[i] = gt_thr[i];
        d->qh[i] = gt_thr[i] == 0.0 ? h_at_zero : (std::nextafter(gt_thr[i], inf) - gt_thr[i]) * 0.5;
    }
    d->qt[3] = p->pswt_1_ndvi;
    d->qh[3] = p->pswt_1_ndvi == 0.0 ? -h_at_zero
                                     : -((p->pswt_1_ndvi - std::nextafter(p->pswt_1_ndvi, -inf)) * 0.5);
    d->awesh4_min = int_gt_min(4.0 * p->awgt);
    d->p1_swir1_max = int_lt_max(p->pswt_1_swir1);
    d->p1_nir_max = int_lt_max(p->pswt_1_nir);
    d->p2_blue_max = int_lt_max(p->pswt_2_blue);
    d->p2_swir1_max = int_lt_max(p->pswt_2_swir1);
    d->p2_swir2_max = int_lt_max(p->pswt_2_swir2);
    d->p2_nir_max = int_lt_max(p->pswt_2_nir);
    d->lc_nir_min = int_gt_min(p->lcmask_nir);
    d->aer_nir_max = int_le_max(p->aerosol_max_nir);
    for (int i = 0; i < 6; ++i) {
        const double f = p->band_fill[i];
        d->band_fill[i] = (std::isfinite(f) && f == std::floor(f) && f >= -32768.0 && f <= 32767.0)
                              ? (int32_t)f : std::numeric_limits<int32_t>::max();
    }
    {
        const double f = p->fmask_fill;
        d->fmask_fill = (std::isfinite(f) && f == std::floor(f) && f >= 0.0 && f <= 255.0) ? (int32_t)f : -1;
    }
    d->clip_min = p->clip_negative_reflectance ? 1 : -32768;
    d->shadow_bits = p->mask_adjacent_to_cloud_mode == DSWX_ADJ_MASK ? (8 | 4) : 8;
    d->collapse = p->collapse_wtr_classes ? 1 : 0;
    {   // _compute_browse_array :3110-3128 applied to each possible uncollapsed WTR code
        const int codes[9] = {0, 1, 2, 3, 4, 252, 253, 254, 255};
        for (int k = 0; k < 9; ++k) {
            int v = codes[k];
            if (p->browse_exclude_psw_aggressive && v == 4) v = 0;
            if (p->collapse_wtr_classes && v <= 4) v = (v + 1) >> 1;
            if (p->browse_not_water_to_nodata && v == 0) v = 255;
            if (p->browse_cloud_to_nodata && v == 253) v = 255;
            if (p->browse_snow_to_nodata && v == 252) v = 255;
            if (p->browse_ocean_masked_to_nodata && v == 254) v = 255;
            d->browse_lut[k >> 2] |= (uint32_t)v << (8 * (k & 3));
        }
    }
    const int cls_of_row[4] = {0, 2, 3, 4};
    for (int v = 0; v < 256 && p->apply_aerosol_class_remapping; ++v) {
        uint32_t bits = 0;
        for (int k = 0; k < 4; ++k)
            if (p->aerosol_fmask_lut[k][v]) bits |= 1u << cls_of_row[k];
        d->aer_lut[v >> 2] |= bits << (8 * (v & 3));
    }
    return DSWX_OK;
}

static uint32_t pack16(int v) { return ((uint32_t)v & 0xffffu) * 0x10001u; }

static void make_lut_consts(const DevParams& d, LutConsts* c) {
    std::memset(c, 0, sizeof *c);
    for (int k = 0; k < 6; ++k) {
        if (d.band_fill[k] == std::numeric_limits<int32_t>::max()) c->fill_off[k] = 0x00010001u;
        else c->fill_pk[k] = pack16(d.band_fill[k]);
    }
    c->clip_pk = pack16(d.clip_min);
    // "x <= k": pack k clamped to int16; below the int16 range the test can never hold
    auto le = [](int32_t k, uint32_t* force) {
        if (k < -32768) { *force = 0x80008000u; return pack16(-32768); }
        return pack16(k > 32767 ? 32767 : k);
    };
    uint32_t f4a = 0, f4b = 0, f5a = 0, f5b = 0, f5c = 0, f5d = 0;
    c->k_p1_swir1 = le(d.p1_swir1_max, &f4a); c->k_p1_nir = le(d.p1_nir_max, &f4b);
    c->k_p2_blue = le(d.p2_blue_max, &f5a); c->k_p2_swir1 = le(d.p2_swir1_max, &f5b);
    c->k_p2_swir2 = le(d.p2_swir2_max, &f5c); c->k_p2_nir = le(d.p2_nir_max, &f5d);
    c->force4 = f4a | f4b; c->force5 = f5a | f5b | f5c | f5d;
    c->k_aer_nir = le(d.aer_nir_max, &c->force_noaer);
    // "nir >= k": above the int16 range never bright, below it always
    if (d.lc_nir_min > 32767) { c->force_dark = 0x80008000u; c->k_lc_nir = pack16(32767); }
    else c->k_lc_nir = pack16(d.lc_nir_min < -32768 ? -32768 : d.lc_nir_min);
    c->awesh_init = -d.awesh4_min;
}

extern "C" {

int dswx_abi_version(void) { return DSWX_ABI_VERSION; }

const char* dswx_last_error(void) { return g_err.c_str(); }

int dswx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dswx_params_default(dswx_params_t* p) {
    if (!p) return fail(DSWX_ERR_ARG, "params is NULL");
    std::memset(p, 0, sizeof *p);
    p->wigt = 0.124; p->awgt = 0.0;
    p->pswt_1_mndwi = -0.44; p->pswt_1_nir = 1500; p->pswt_1_swir1 = 900; p->pswt_1_ndvi = 0.7;
    p->pswt_2_mndwi = -0.5; p->pswt_2_blue = 1000; p->pswt_2_nir = 2500; p->pswt_2_swir1 = 3000;
    p->pswt_2_swir2 = 1000; p->lcmask_nir = 1200;
    for (int i = 0; i < 6; ++i) p->band_fill[i] = -9999.0;
    p->fmask_fill = 255.0;
    p->aerosol_max_nir = 0.1 / 0.0001;
    p->clip_negative_reflectance = 1;
    p->mask_adjacent_to_cloud_mode = DSWX_ADJ_MASK;
    p->apply_aerosol_class_remapping = 1;
    p->collapse_wtr_classes = 1;
    // defaults/dswx_hls.yaml:128-169 (browse_image_group) and :5316
    p->browse_exclude_psw_aggressive = 1;
    p->browse_ocean_masked_to_nodata = 1;
    const int l3[] = {224, 160, 96}, l5[] = {224, 192, 160, 128, 96};
    for (int v : l3) { p->aerosol_fmask_lut[0][v] = 1; p->aerosol_fmask_lut[1][v] = 1; }
    for (int v : l5) { p->aerosol_fmask_lut[2][v] = 1; p->aerosol_fmask_lut[3][v] = 1; }
    return DSWX_OK;
}

int dswx_ctx_create(int device, dswx_ctx_t** out) {
    if (!out) return fail(DSWX_ERR_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(DSWX_ERR_NO_DEVICE, "no HIP device visible: the DSWx HIP path has no CPU fallback");
    if (device < 0 || device >= n) return fail(DSWX_ERR_ARG, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    dswx_ctx* c = new dswx_ctx();
    c->device = device;
    if (const char* e = std::getenv("DSWX_FUSED_VARIANT")) {
        const int v = std::atoi(e);
        c->fused_variant = (v >= 1 && v <= 4) ? v : 0;
    }
    if (const char* e = std::getenv("DSWX_TUNE_WPS")) c->tune_wps = std::atoi(e);
    if (const char* e = std::getenv("DSWX_TUNE_CHUNKS")) c->tune_chunks = std::atoi(e);
    if (const char* e = std::getenv("DSWX_TUNE_ABLATE")) c->tune_ablate = std::atoi(e);
    if (const char* e = std::getenv("DSWX_TUNE_LUT_WPS")) c->tune_lut_wps = std::atoi(e);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(DSWX_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return DSWX_OK;
}

int dswx_ctx_destroy(dswx_ctx_t* ctx) {
    if (!ctx) return DSWX_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->cover) (void)hipFree(ctx->cover);
    if (ctx->tables) (void)hipFree(ctx->tables);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return DSWX_OK;
}

static bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// height/width are only needed (and only trusted) in 'cover' mode; 0 = unknown
static int classify_device_impl(dswx_ctx_t* ctx, const dswx_params_t* params, int64_t n_tiles, int64_t n_pixels,
                                int64_t height, int64_t width, const dswx_planes_in_t* in,
                                const dswx_planes_out_t* out, int64_t* counters, void* stream) {
    if (!ctx || !params || !in || !out) return fail(DSWX_ERR_ARG, "NULL argument");
    if (n_tiles < 0 || n_pixels < 0) return fail(DSWX_ERR_ARG, "negative size");
    for (int k = 0; k < 6; ++k)
        if (!in->band[k]) return fail(DSWX_ERR_ARG, "band[%d] is NULL", k);
    if (!in->fmask) return fail(DSWX_ERR_ARG, "fmask is NULL");
    KArgs a;
    int rc = make_dev_params(params, &a.P);
    if (rc) return rc;
    const bool cover = params->mask_adjacent_to_cloud_mode == DSWX_ADJ_COVER;
    if (cover && (height <= 0 || width <= 0 || height * width != n_pixels))
        return fail(DSWX_ERR_UNSUPPORTED,
                    "mask_adjacent_to_cloud_mode 'cover' is a 2-D neighbourhood operation: use "
                    "dswx_classify_device_2d / dswx_classify_host, which know the tile height and width");
    if (cover && (height > 2147483647LL || width > 2147483647LL)) return fail(DSWX_ERR_ARG, "tile too large");
    for (int k = 0; k < 6; ++k)
        if (!aligned_to(in->band[k], 2)) return fail(DSWX_ERR_ALIGN, "band[%d] not 2-byte aligned", k);
    if (out->diag && !aligned_to(out->diag, 2)) return fail(DSWX_ERR_ALIGN, "diag not 2-byte aligned");
    if (counters && !aligned_to(counters, 8)) return fail(DSWX_ERR_ALIGN, "counters not 8-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (counters && n_tiles > 0)
        HIP_TRY(hipMemsetAsync(counters, 0, (size_t)n_tiles * 3 * sizeof(int64_t), s));
    if (n_tiles == 0 || n_pixels == 0) { ctx->last_kernel = "none (empty input)"; return DSWX_OK; }
    a.in = *in;
    a.out = *out;
    a.counters = reinterpret_cast<unsigned long long*>(counters);
    a.n_pixels = n_pixels;
    a.cover_w2 = a.cover_pc = nullptr;
    a.height = (int)height; a.width = (int)width;
    dswx_planes_out_t final_out = *out;     // what stage 2 of 'cover' writes
    if (cover) {
        const size_t need = 2 * (size_t)n_tiles * (size_t)n_pixels;
        if (need > ctx->cover_bytes) {
            HIP_TRY(hipStreamSynchronize(s));
            if (ctx->cover) HIP_TRY(hipFree(ctx->cover));
            ctx->cover = nullptr; ctx->cover_bytes = 0;
            HIP_TRY(hipMalloc(&ctx->cover, need));
            ctx->cover_bytes = need;
        }
        a.cover_w2 = static_cast<uint8_t*>(ctx->cover);
        a.cover_pc = a.cover_w2 + (size_t)n_tiles * (size_t)n_pixels;
        // stage 1 stops before the snow step: these four layers come from stage 2
        a.out.wtr = a.out.bwtr = a.out.conf = a.out.cloud = a.out.browse = nullptr;
    }

    const bool any_index = out->mndwi || out->ndvi || out->awesh;
    const bool masks = in->land || in->shad || in->ocean;
    // the fused kernel needs every plane 16-byte aligned at every tile start
    bool vec_ok = (n_pixels % 16 == 0) || n_tiles == 1;
    for (int k = 0; k < 6 && vec_ok; ++k) vec_ok = aligned_to(in->band[k], 16);
    vec_ok = vec_ok && aligned_to(in->fmask, 16) && (!in->land || aligned_to(in->land, 16)) &&
             (!in->shad || aligned_to(in->shad, 16)) && (!in->ocean || aligned_to(in->ocean, 16)) &&
             (!out->diag || aligned_to(out->diag, 16));
    uint8_t* const u8outs[] = {out->wtr1, out->wtr1_aerosol, out->wtr2, out->wtr, out->bwtr, out->conf, out->cloud,
                               out->browse};
    for (uint8_t* p : u8outs) vec_ok = vec_ok && (!p || aligned_to(p, 16));

    const int64_t max_y = 65535;
    char info[256];
    for (int64_t t0 = 0; t0 < n_tiles; t0 += max_y) {
        const int64_t nt = (n_tiles - t0 < max_y) ? n_tiles - t0 : max_y;
        KArgs b = a;
        const int64_t shift = t0 * n_pixels;
        for (int k = 0; k < 6; ++k) b.in.band[k] += shift;
        b.in.fmask += shift;
        if (b.in.land) b.in.land += shift;
        if (b.in.shad) b.in.shad += shift;
        if (b.in.ocean) b.in.ocean += shift;
        if (b.out.diag) b.out.diag += shift;
        if (b.out.wtr1) b.out.wtr1 += shift;
        if (b.out.wtr1_aerosol) b.out.wtr1_aerosol += shift;
        if (b.out.wtr2) b.out.wtr2 += shift;
        if (b.out.wtr) b.out.wtr += shift;
        if (b.out.bwtr) b.out.bwtr += shift;
        if (b.out.conf) b.out.conf += shift;
        if (b.out.cloud) b.out.cloud += shift;
        if (b.out.browse) b.out.browse += shift;
        if (b.out.mndwi) b.out.mndwi += shift;
        if (b.out.ndvi) b.out.ndvi += shift;
        if (b.out.awesh) b.out.awesh += shift;
        if (b.counters) b.counters += t0 * 3;
        if (b.cover_w2) { b.cover_w2 += shift; b.cover_pc += shift; }
        b.px_begin = 0;
        b.partials = nullptr;
        const int64_t groups = vec_ok ? (n_pixels >> 3) : 0;
        if (groups > 0) {
            // 'cover' stage 1 and the browse plane live in the direct kernel only
            const bool plain_outputs = !cover && !b.out.browse;
            const bool staged = ctx->fused_variant == 1 && plain_outputs;
            const bool wspec = ctx->fused_variant == 2 && plain_outputs;
            const bool tabled = ctx->fused_variant == 3 && plain_outputs;
            const bool wslut = ctx->fused_variant == 4 && plain_outputs;
            const int threads = staged ? FUSED_THREADS : 256;
            const int lut_chunks = (ctx->tune_chunks == 1 || ctx->tune_chunks == 4) ? ctx->tune_chunks : 1;
            const int64_t per_block = (int64_t)threads * (tabled ? lut_chunks : 1);
            const int64_t gx = (groups + per_block - 1) / per_block;
            const int waves = threads / 64;
            dim3 grid((unsigned)gx, (unsigned)nt), block(threads);
            if (b.counters) {
                const size_t need = (size_t)nt * (size_t)gx * waves * sizeof(uint2);
                if (need > ctx->partials_bytes) {
                    HIP_TRY(hipStreamSynchronize(s));
                    if (ctx->partials) HIP_TRY(hipFree(ctx->partials));
                    ctx->partials = nullptr; ctx->partials_bytes = 0;
                    HIP_TRY(hipMalloc(&ctx->partials, need));
                    ctx->partials_bytes = need;
                }
                b.partials = static_cast<uint2*>(ctx->partials);
            }
            uint8_t* const u8p[7] = {b.out.wtr1, b.out.wtr1_aerosol, b.out.wtr2, b.out.wtr, b.out.bwtr, b.out.conf, b.out.cloud};
            b.n_u8_out = 0;
            for (int i = 0; i < 7; ++i)
                if (u8p[i]) { b.u8_out[b.n_u8_out] = u8p[i]; b.u8_region[b.n_u8_out] = i; ++b.n_u8_out; }
            b.n_diag_pieces = b.out.diag ? 8 : 0;
            if (wslut) {
                if (!ctx->tables) HIP_TRY(hipMalloc(&ctx->tables, sizeof(Tables)));
                Tables* tabs = static_cast<Tables*>(ctx->tables);
                LutConsts lc;
                make_lut_consts(b.P, &lc);
                hipLaunchKernelGGL(dswx_build_tables, dim3(4), dim3(256), 0, s, b.P, tabs);
                const int wps = ctx->tune_lut_wps;
#define WSLUT_LAUNCH(M, W) hipLaunchKernelGGL((dswx_classify_wslut<M, W>), grid, block, 0, s, b, lc, tabs)
                if (masks) { if (wps >= 5) WSLUT_LAUNCH(true, 5); else if (wps == 4) WSLUT_LAUNCH(true, 4); else WSLUT_LAUNCH(true, 3); }
                else if (ctx->tune_ablate == 1) hipLaunchKernelGGL((dswx_classify_wslut<false, 4, 1>), grid, block, 0, s, b, lc, tabs);
                else if (ctx->tune_ablate == 2) hipLaunchKernelGGL((dswx_classify_wslut<false, 4, 2>), grid, block, 0, s, b, lc, tabs);
                else if (ctx->tune_ablate == 3) hipLaunchKernelGGL((dswx_classify_wslut<false, 4, 3>), grid, block, 0, s, b, lc, tabs);
                else { if (wps >= 5) WSLUT_LAUNCH(false, 5); else if (wps == 4) WSLUT_LAUNCH(false, 4); else WSLUT_LAUNCH(false, 3); }
                snprintf(info, sizeof info, "dswx_classify_wslut<%s> (warp-specialised + table-driven) grid=(%lld,%lld) block=256 wps=%d",
                         masks ? "true" : "false", (long long)gx, (long long)nt, wps);
            } else if (tabled) {
                if (!ctx->tables) HIP_TRY(hipMalloc(&ctx->tables, sizeof(Tables)));
                Tables* tabs = static_cast<Tables*>(ctx->tables);
                LutConsts lc;
                make_lut_consts(b.P, &lc);
                hipLaunchKernelGGL(dswx_build_tables, dim3(4), dim3(256), 0, s, b.P, tabs);
                const int wps = ctx->tune_lut_wps;
#define LUT_LAUNCH(M, CH, W) hipLaunchKernelGGL((dswx_classify_lut<M, CH, W>), grid, block, 0, s, b, lc, tabs)
#define LUT_SEL_W(M, CH) do { if (wps >= 6) LUT_LAUNCH(M, CH, 6); else if (wps == 5) LUT_LAUNCH(M, CH, 5); else LUT_LAUNCH(M, CH, 4); } while (0)
#define LUT_SEL_C(M) do { if (lut_chunks == 4) LUT_SEL_W(M, 4); else LUT_SEL_W(M, 1); } while (0)
                if (masks) LUT_SEL_C(true); else LUT_SEL_C(false);
                snprintf(info, sizeof info, "dswx_classify_lut<%s> (table-driven) grid=(%lld,%lld) block=256 chunks=%d wps=%d",
                         masks ? "true" : "false", (long long)gx, (long long)nt, lut_chunks, wps);
            } else if (wspec) {
                if (masks) hipLaunchKernelGGL(dswx_classify_ws<true>, grid, block, 0, s, b);
                else hipLaunchKernelGGL(dswx_classify_ws<false>, grid, block, 0, s, b);
                snprintf(info, sizeof info, "dswx_classify_ws<%s> (warp-specialised, LDS-DMA) grid=(%lld,%lld) block=256",
                         masks ? "true" : "false", (long long)gx, (long long)nt);
            } else if (staged) {
                if (masks) hipLaunchKernelGGL(dswx_classify_fused<true>, grid, block, 0, s, b);
                else hipLaunchKernelGGL(dswx_classify_fused<false>, grid, block, 0, s, b);
                snprintf(info, sizeof info, "dswx_classify_fused<%s> (LDS-staged) grid=(%lld,%lld) block=%d lds=%d",
                         masks ? "true" : "false", (long long)gx, (long long)nt, FUSED_THREADS, STAGE_BYTES);
            } else {
                const bool extras = b.out.browse || b.cover_w2;
                if (masks && extras) hipLaunchKernelGGL((dswx_classify_v8<true, true>), grid, block, 0, s, b);
                else if (masks) hipLaunchKernelGGL((dswx_classify_v8<true, false>), grid, block, 0, s, b);
                else if (extras) hipLaunchKernelGGL((dswx_classify_v8<false, true>), grid, block, 0, s, b);
                else if (ctx->tune_wps == 6) hipLaunchKernelGGL((dswx_classify_v8<false, false, 6>), grid, block, 0, s, b);
                else if (ctx->tune_wps == 8) hipLaunchKernelGGL((dswx_classify_v8<false, false, 8>), grid, block, 0, s, b);
                else hipLaunchKernelGGL((dswx_classify_v8<false, false, 4>), grid, block, 0, s, b);
                snprintf(info, sizeof info, "dswx_classify_v8<%s,%s> (fused, direct stores) grid=(%lld,%lld) block=256",
                         masks ? "true" : "false", extras ? "true" : "false", (long long)gx, (long long)nt);
            }
            HIP_TRY(hipGetLastError());
            if (b.counters) {
                hipLaunchKernelGGL(dswx_counters_finish, dim3((unsigned)nt), dim3(256), 0, s, b.partials,
                                   b.counters, (long long)gx * waves, in->ocean ? 1 : 0, (long long)groups * 8);
                HIP_TRY(hipGetLastError());
            }
            b.px_begin = groups * 8;
        }
        if (b.px_begin < n_pixels) {
            const int64_t rest = n_pixels - b.px_begin;
            const int64_t gx = (rest + 255) / 256;
            dim3 grid((unsigned)gx, (unsigned)nt), block(256);
            hipLaunchKernelGGL(dswx_classify_v1, grid, block, 0, s, b);
            if (groups == 0)
                snprintf(info, sizeof info, "dswx_classify_v1 grid=(%lld,%lld) block=256",
                         (long long)gx, (long long)nt);
        }
        HIP_TRY(hipGetLastError());
        if (cover) {
            KArgs c2 = b;
            c2.out = final_out;
            if (c2.out.wtr) c2.out.wtr += shift;
            if (c2.out.bwtr) c2.out.bwtr += shift;
            if (c2.out.conf) c2.out.conf += shift;
            if (c2.out.cloud) c2.out.cloud += shift;
            if (c2.out.browse) c2.out.browse += shift;
            if (c2.out.wtr || c2.out.bwtr || c2.out.conf || c2.out.cloud || c2.out.browse) {
                dim3 grid((unsigned)((width + CV_TILE - 1) / CV_TILE), (unsigned)((height + CV_TILE - 1) / CV_TILE),
                          (unsigned)nt);
                hipLaunchKernelGGL(dswx_cover_stage2, grid, dim3(256), 0, s, c2);
                HIP_TRY(hipGetLastError());
            }
            const size_t len = strlen(info);
            snprintf(info + len, sizeof info - len, " + dswx_cover_stage2 grid=(%lld,%lld,%lld)",
                     (long long)((width + CV_TILE - 1) / CV_TILE), (long long)((height + CV_TILE - 1) / CV_TILE),
                     (long long)nt);
        }
        if (any_index) {
            const long long total = (long long)nt * n_pixels;
            dim3 grid((unsigned)((total + 255) / 256)), block(256);
            hipLaunchKernelGGL(dswx_indices_v1, grid, block, 0, s, b, total);
            HIP_TRY(hipGetLastError());
        }
    }
    ctx->last_kernel = info;
    return DSWX_OK;
}

int dswx_classify_device(dswx_ctx_t* ctx, const dswx_params_t* params, int64_t n_tiles, int64_t n_pixels,
                         const dswx_planes_in_t* in, const dswx_planes_out_t* out, int64_t* counters,
                         void* stream) {
    return classify_device_impl(ctx, params, n_tiles, n_pixels, 0, 0, in, out, counters, stream);
}

int dswx_classify_device_2d(dswx_ctx_t* ctx, const dswx_params_t* params, int64_t n_tiles, int64_t height,
                            int64_t width, const dswx_planes_in_t* in, const dswx_planes_out_t* out,
                            int64_t* counters, void* stream) {
    if (height < 0 || width < 0) return fail(DSWX_ERR_ARG, "negative size");
    return classify_device_impl(ctx, params, n_tiles, height * width, height, width, in, out, counters, stream);
}

int dswx_classify_host(dswx_ctx_t* ctx, const dswx_params_t* params, int64_t n_tiles, int64_t height,
                       int64_t width, const dswx_planes_in_t* in, const dswx_planes_out_t* out,
                       int64_t* counters) {
    if (!ctx || !params || !in || !out) return fail(DSWX_ERR_ARG, "NULL argument");
    if (n_tiles < 0 || height < 0 || width < 0) return fail(DSWX_ERR_ARG, "negative size");
    for (int k = 0; k < 6; ++k)
        if (!in->band[k]) return fail(DSWX_ERR_ARG, "band[%d] is NULL", k);
    if (!in->fmask) return fail(DSWX_ERR_ARG, "fmask is NULL");
    {   // validate parameters before touching the device
        DevParams tmp;
        int rc = make_dev_params(params, &tmp);
        if (rc) return rc;
    }
    const int64_t P = height * width;
    if (n_tiles == 0 || P == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    // one tile at a time through a grow-only device arena: planes at 256-byte
    // aligned offsets so the vector kernel is always eligible
    auto rnd = [](size_t x) { return (x + 255) & ~size_t(255); };
    size_t off = 0;
    size_t o_band[6], o_fm, o_land = 0, o_shad = 0, o_ocean = 0;
    for (int k = 0; k < 6; ++k) { o_band[k] = off; off += rnd((size_t)P * 2); }
    o_fm = off; off += rnd((size_t)P);
    if (in->land) { o_land = off; off += rnd((size_t)P); }
    if (in->shad) { o_shad = off; off += rnd((size_t)P); }
    if (in->ocean) { o_ocean = off; off += rnd((size_t)P); }
    size_t o_diag = off; if (out->diag) off += rnd((size_t)P * 2);
    uint8_t* const h_u8[8] = {out->wtr1, out->wtr1_aerosol, out->wtr2, out->wtr, out->bwtr, out->conf, out->cloud,
                              out->browse};
    size_t o_u8[8];
    for (int i = 0; i < 8; ++i) { o_u8[i] = off; if (h_u8[i]) off += rnd((size_t)P); }
    double* const h_f64[3] = {out->mndwi, out->ndvi, out->awesh};
    size_t o_f64[3];
    for (int i = 0; i < 3; ++i) { o_f64[i] = off; if (h_f64[i]) off += rnd((size_t)P * 8); }
    size_t o_cnt = off; off += 256;
    if (off > ctx->stage_bytes) {
        if (ctx->stage) HIP_TRY(hipFree(ctx->stage));
        ctx->stage = nullptr; ctx->stage_bytes = 0;
        HIP_TRY(hipMalloc(&ctx->stage, off));
        ctx->stage_bytes = off;
    }
    char* base = static_cast<char*>(ctx->stage);
    hipStream_t s = ctx->stream;
    for (int64_t t = 0; t < n_tiles; ++t) {
        const size_t sh = (size_t)t * (size_t)P;
        dswx_planes_in_t din{};
        dswx_planes_out_t dout{};
        for (int k = 0; k < 6; ++k) {
            HIP_TRY(hipMemcpyAsync(base + o_band[k], in->band[k] + sh, (size_t)P * 2, hipMemcpyHostToDevice, s));
            din.band[k] = reinterpret_cast<const int16_t*>(base + o_band[k]);
        }
        HIP_TRY(hipMemcpyAsync(base + o_fm, in->fmask + sh, (size_t)P, hipMemcpyHostToDevice, s));
        din.fmask = reinterpret_cast<const uint8_t*>(base + o_fm);
        if (in->land) { HIP_TRY(hipMemcpyAsync(base + o_land, in->land + sh, (size_t)P, hipMemcpyHostToDevice, s)); din.land = reinterpret_cast<const uint8_t*>(base + o_land); }
        if (in->shad) { HIP_TRY(hipMemcpyAsync(base + o_shad, in->shad + sh, (size_t)P, hipMemcpyHostToDevice, s)); din.shad = reinterpret_cast<const uint8_t*>(base + o_shad); }
        if (in->ocean) { HIP_TRY(hipMemcpyAsync(base + o_ocean, in->ocean + sh, (size_t)P, hipMemcpyHostToDevice, s)); din.ocean = reinterpret_cast<const uint8_t*>(base + o_ocean); }
        if (out->diag) dout.diag = reinterpret_cast<uint16_t*>(base + o_diag);
        uint8_t** const d_u8[8] = {&dout.wtr1, &dout.wtr1_aerosol, &dout.wtr2, &dout.wtr, &dout.bwtr, &dout.conf, &dout.cloud,
                                   &dout.browse};
        for (int i = 0; i < 8; ++i) if (h_u8[i]) *d_u8[i] = reinterpret_cast<uint8_t*>(base + o_u8[i]);
        double** const d_f64[3] = {&dout.mndwi, &dout.ndvi, &dout.awesh};
        for (int i = 0; i < 3; ++i) if (h_f64[i]) *d_f64[i] = reinterpret_cast<double*>(base + o_f64[i]);
        int64_t* dcnt = counters ? reinterpret_cast<int64_t*>(base + o_cnt) : nullptr;
        int rc = dswx_classify_device_2d(ctx, params, 1, height, width, &din, &dout, dcnt, s);
        if (rc) return rc;
        if (out->diag) HIP_TRY(hipMemcpyAsync(out->diag + sh, dout.diag, (size_t)P * 2, hipMemcpyDeviceToHost, s));
        for (int i = 0; i < 8; ++i)
            if (h_u8[i]) HIP_TRY(hipMemcpyAsync(h_u8[i] + sh, *d_u8[i], (size_t)P, hipMemcpyDeviceToHost, s));
        for (int i = 0; i < 3; ++i)
            if (h_f64[i]) HIP_TRY(hipMemcpyAsync(h_f64[i] + sh, *d_f64[i], (size_t)P * 8, hipMemcpyDeviceToHost, s));
        if (counters) HIP_TRY(hipMemcpyAsync(counters + t * 3, dcnt, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return DSWX_OK;
}

int dswx_interpret_layer_host(dswx_ctx_t* ctx, const int64_t* diag_decimal, int64_t n, uint8_t* out) {
    if (!ctx || (n > 0 && (!diag_decimal || !out))) return fail(DSWX_ERR_ARG, "NULL argument");
    if (n < 0) return fail(DSWX_ERR_ARG, "negative size");
    if (n == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    void* d_in = nullptr;
    void* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_in, (size_t)n * 8));
    hipError_t e = hipMalloc(&d_out, (size_t)n);
    if (e != hipSuccess) { (void)hipFree(d_in); return fail(DSWX_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e)); }
    hipStream_t s = ctx->stream;
    e = hipMemcpyAsync(d_in, diag_decimal, (size_t)n * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(dswx_interpret_v1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                           static_cast<const long long*>(d_in), static_cast<uint8_t*>(d_out), (long long)n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(DSWX_ERR_HIP, "dswx_interpret_layer_host: %s", hipGetErrorString(e));
    return DSWX_OK;
}

static int shadow_args(ShadowArgs* a, int64_t height, int64_t width, int64_t margin, const double sun_vector[3],
                       double sin_azimuth, double cos_azimuth, double min_slope_angle,
                       double max_sun_local_inc_angle, double pixel_spacing_x, double pixel_spacing_y) {
    if (!sun_vector) return fail(DSWX_ERR_ARG, "sun_vector is NULL");
    if (height < 2 || width < 2)
        return fail(DSWX_ERR_ARG, "Shape of array too small to calculate a numerical gradient, "
                                  "at least 2 elements are required.");
    if (margin < 0 || 2 * margin >= height || 2 * margin >= width) return fail(DSWX_ERR_ARG, "bad margin");
    a->height = height; a->width = width; a->margin = margin;
    a->spacing_x = (float)pixel_spacing_x;
    a->neg_abs_spacing_y = (float)(-std::fabs(pixel_spacing_y));
    for (int i = 0; i < 3; ++i) a->sun[i] = sun_vector[i];
    a->sin_az = sin_azimuth; a->cos_az = cos_azimuth;
    a->min_slope_angle = min_slope_angle; a->max_sun_local_inc_angle = max_sun_local_inc_angle;
    return DSWX_OK;
}

int dswx_shadow_layer_device(dswx_ctx_t* ctx, const float* dem, int64_t n_tiles, int64_t height, int64_t width,
                             int64_t margin, const double sun_vector[3], double sin_azimuth,
                             double cos_azimuth, double min_slope_angle, double max_sun_local_inc_angle,
                             double pixel_spacing_x, double pixel_spacing_y, uint8_t* shadow, void* stream) {
    if (!ctx || !dem || !shadow) return fail(DSWX_ERR_ARG, "NULL argument");
    if (n_tiles < 0 || n_tiles > 65535) return fail(DSWX_ERR_ARG, "n_tiles out of range");
    ShadowArgs a;
    int rc = shadow_args(&a, height, width, margin, sun_vector, sin_azimuth, cos_azimuth, min_slope_angle,
                         max_sun_local_inc_angle, pixel_spacing_x, pixel_spacing_y);
    if (rc) return rc;
    if (n_tiles == 0) return DSWX_OK;
    a.dem = dem; a.shadow = shadow;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const long long ow = width - 2 * margin, oh = height - 2 * margin;
    dim3 grid((unsigned)((ow + 63) / 64), (unsigned)((oh + 3) / 4), (unsigned)n_tiles), block(256);
    if (grid.y > 65535) return fail(DSWX_ERR_ARG, "raster too tall for one launch");
    hipLaunchKernelGGL(dswx_shadow_v1, grid, block, 0, s, a);
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

int dswx_shadow_layer_host(dswx_ctx_t* ctx, const float* dem, int64_t height, int64_t width, int64_t margin,
                           const double sun_vector[3], double sin_azimuth, double cos_azimuth,
                           double min_slope_angle, double max_sun_local_inc_angle, double pixel_spacing_x,
                           double pixel_spacing_y, uint8_t* shadow) {
    if (!ctx || !dem || !shadow) return fail(DSWX_ERR_ARG, "NULL argument");
    ShadowArgs chk;
    int rc = shadow_args(&chk, height, width, margin, sun_vector, sin_azimuth, cos_azimuth, min_slope_angle,
                         max_sun_local_inc_angle, pixel_spacing_x, pixel_spacing_y);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t in_bytes = (size_t)height * (size_t)width * 4;
    const size_t out_px = (size_t)(height - 2 * margin) * (size_t)(width - 2 * margin);
    void* d_dem = nullptr;
    void* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_dem, in_bytes));
    hipError_t e = hipMalloc(&d_out, out_px);
    hipStream_t s = ctx->stream;
    if (e == hipSuccess) e = hipMemcpyAsync(d_dem, dem, in_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        rc = dswx_shadow_layer_device(ctx, static_cast<const float*>(d_dem), 1, height, width, margin, sun_vector,
                                      sin_azimuth, cos_azimuth, min_slope_angle, max_sun_local_inc_angle,
                                      pixel_spacing_x, pixel_spacing_y, static_cast<uint8_t*>(d_out), s);
        if (rc == DSWX_OK) e = hipMemcpyAsync(shadow, d_out, out_px, hipMemcpyDeviceToHost, s);
        if (rc == DSWX_OK && e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void)hipFree(d_dem);
    if (d_out) (void)hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) return fail(DSWX_ERR_HIP, "dswx_shadow_layer_host: %s", hipGetErrorString(e));
    return DSWX_OK;
}

int dswx_landcover_mask_host(dswx_ctx_t* ctx, const uint8_t* worldcover_up3, const uint8_t* copernicus,
                             int64_t height, int64_t width, const int32_t* forest_classes,
                             int32_t n_forest_classes, const int32_t thresholds[4], int32_t year_offset,
                             uint8_t* land) {
    if (!ctx || !worldcover_up3 || !copernicus || !thresholds || !land) return fail(DSWX_ERR_ARG, "NULL argument");
    if (height < 0 || width < 0 || n_forest_classes < 0 || (n_forest_classes > 0 && !forest_classes))
        return fail(DSWX_ERR_ARG, "bad size");
    if (height == 0 || width == 0) return DSWX_OK;
    LandArgs a;
    std::memset(&a, 0, sizeof a);
    for (int i = 0; i < n_forest_classes; ++i) {
        const int c = forest_classes[i];
        if (c >= 0 && c <= 255) a.forest_bits[c >> 5] |= 1u << (c & 31);
    }
    a.thr_tree = thresholds[0]; a.thr_low = thresholds[1]; a.thr_high = thresholds[2]; a.thr_water = thresholds[3];
    // numpy stores the class through a uint8 array: values wrap modulo 256
    a.low_class = (int)(uint8_t)(0 + year_offset);
    a.high_class = (int)(uint8_t)(100 + year_offset);
    a.height = height; a.width = width;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)height * (size_t)width;
    void* d_wc = nullptr; void* d_cg = nullptr; void* d_out = nullptr;
    hipError_t e = hipMalloc(&d_wc, 9 * n);
    if (e == hipSuccess) e = hipMalloc(&d_cg, n);
    if (e == hipSuccess) e = hipMalloc(&d_out, n);
    hipStream_t s = ctx->stream;
    if (e == hipSuccess) e = hipMemcpyAsync(d_wc, worldcover_up3, 9 * n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cg, copernicus, n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        a.wc3 = static_cast<const uint8_t*>(d_wc); a.cgls = static_cast<const uint8_t*>(d_cg);
        a.land = static_cast<uint8_t*>(d_out);
        dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4)), block(256);
        hipLaunchKernelGGL(dswx_landcover_v1, grid, block, 0, s, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(land, d_out, n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (d_wc) (void)hipFree(d_wc);
    if (d_cg) (void)hipFree(d_cg);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(DSWX_ERR_HIP, "dswx_landcover_mask_host: %s", hipGetErrorString(e));
    return DSWX_OK;
}

int dswx_stream_probe(dswx_ctx_t* ctx, int64_t n_tiles, int64_t n_pixels, const dswx_planes_in_t* in,
                      const dswx_planes_out_t* out, int variant, void* stream) {
    if (!ctx || !in || !out) return fail(DSWX_ERR_ARG, "NULL argument");
    if (n_tiles <= 0 || n_tiles > 65535 || n_pixels <= 0 || n_pixels % 16)
        return fail(DSWX_ERR_ARG, "probe needs 1..65535 tiles of a multiple of 16 pixels");
    if (!out->diag || !out->wtr1 || !out->wtr2 || !out->wtr || !out->bwtr || !out->conf || !out->cloud || !in->fmask)
        return fail(DSWX_ERR_ARG, "probe needs all seven output planes");
    for (int k = 0; k < 6; ++k)
        if (!in->band[k] || !aligned_to(in->band[k], 16)) return fail(DSWX_ERR_ALIGN, "band[%d]", k);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    KArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = *in; a.out = *out; a.n_pixels = n_pixels;
    // variant = ppt16 | nt << 1 | log2(iters) << 2 | mode << 9 | xcdmap << 11 |
    // block512 << 12 ; bit 8: flat two-stream copy of
    // the same byte counts (needs the planes laid out as DeviceBatch does:
    // band[0..5], fmask contiguous; diag, wtr1.. contiguous)
    if (variant & 256) {
        const long long total = n_tiles * n_pixels;
        const long long n16_in = total * 13 / 16, n16_out = total * 8 / 16;
        dim3 grid((unsigned)((n16_in + 255) / 256)), block(256);
        if (variant & 2) hipLaunchKernelGGL(dswx_flat_copy_k<true>, grid, block, 0, s, (const u32x4*)in->band[0], (u32x4*)out->diag, n16_in, n16_out);
        else hipLaunchKernelGGL(dswx_flat_copy_k<false>, grid, block, 0, s, (const u32x4*)in->band[0], (u32x4*)out->diag, n16_in, n16_out);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & 524288) {  // warp-specialised LDS-DMA data movement, bit 1 = nt
        dim3 grid((unsigned)(n_pixels / 2048), (unsigned)n_tiles), block(256);
        if (variant & 2) hipLaunchKernelGGL(dswx_ws_probe_k<true>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(dswx_ws_probe_k<false>, grid, block, 0, s, a);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & 262144) {  // plane-specialised waves, bit 1 = nt
        const long long total = n_tiles * n_pixels;
        dim3 grid((unsigned)(total / 4096)), block(448);
        if (variant & 2) hipLaunchKernelGGL(dswx_plane_per_wave_k<true>, grid, block, 0, s, a, total);
        else hipLaunchKernelGGL(dswx_plane_per_wave_k<false>, grid, block, 0, s, a, total);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & 131072) {  // chunk-interleaved layout: bits 2-3 select CH = 4096 << (4*sel), bit 1 = nt
        const long long total = n_tiles * n_pixels;
        const int sel = (variant >> 2) & 3;
        dim3 grid((unsigned)((total / 8 + 255) / 256)), block(256);
        uint8_t* arena = const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(in->band[0]));
        const bool wnt = variant & 2;
#define CHUNK_LAUNCH(CH) do { if (wnt) hipLaunchKernelGGL((dswx_chunked_layout_probe_k<CH, true>), grid, block, 0, s, arena, total); else hipLaunchKernelGGL((dswx_chunked_layout_probe_k<CH, false>), grid, block, 0, s, arena, total); } while (0)
        if (sel == 0) CHUNK_LAUNCH(4096); else if (sel == 1) CHUNK_LAUNCH(65536); else if (sel == 2) CHUNK_LAUNCH(1048576); else CHUNK_LAUNCH(16777216);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & 65536) {  // role split: bit 2 = SPLIT, bit 1 = nt
        const int64_t groups = n_pixels >> 3;
        const bool wnt = variant & 2;
        if (variant & 8) {
            dim3 grid((unsigned)((((groups + 511) / 512 + 7) / 8) * 16), (unsigned)n_tiles), block(256);
            if (wnt) hipLaunchKernelGGL((dswx_role_split_k<2, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((dswx_role_split_k<2, false>), grid, block, 0, s, a);
        } else if (variant & 4) {
            dim3 grid((unsigned)((groups + 255) / 256), (unsigned)n_tiles), block(256);
            if (wnt) hipLaunchKernelGGL((dswx_role_split_k<1, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((dswx_role_split_k<1, false>), grid, block, 0, s, a);
        } else {
            dim3 grid((unsigned)((((groups + 511) / 512 + 7) / 8) * 16), (unsigned)n_tiles), block(256);
            if (wnt) hipLaunchKernelGGL((dswx_role_split_k<0, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((dswx_role_split_k<0, false>), grid, block, 0, s, a);
        }
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & 32768) {  // plane-per-block stream-count calibration, bit 1 = nt
        const long long total = n_tiles * n_pixels;
        dim3 grid((unsigned)(((total + 4095) / 4096) * 14)), block(256);
        if (variant & 2) hipLaunchKernelGGL(dswx_plane_per_block_k<true>, grid, block, 0, s, a, total);
        else hipLaunchKernelGGL(dswx_plane_per_block_k<false>, grid, block, 0, s, a, total);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & 16384) {  // staged probe: bits 2-3 = log2(BLOCK/256), bit 1 = nt
        const int lb = (variant >> 2) & 3;
        const int bs = 256 << lb;
        const int64_t groups = n_pixels >> 3;
        dim3 grid((unsigned)((groups + bs - 1) / bs), (unsigned)n_tiles), block(bs);
        const bool wnt = variant & 2;
        if (lb == 0) { if (wnt) hipLaunchKernelGGL((dswx_staged_probe_k<256, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((dswx_staged_probe_k<256, false>), grid, block, 0, s, a); }
        else if (lb == 1) { if (wnt) hipLaunchKernelGGL((dswx_staged_probe_k<512, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((dswx_staged_probe_k<512, false>), grid, block, 0, s, a); }
        else { if (wnt) hipLaunchKernelGGL((dswx_staged_probe_k<1024, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((dswx_staged_probe_k<1024, false>), grid, block, 0, s, a); }
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & 8192) {   // write-path calibration: bits 2-3 = WMODE, bit 1 = nt
        const long long total = n_tiles * n_pixels;
        const int wm = (variant >> 2) & 3;
        const bool wnt = variant & 2;
        if (wm == 0) {
            dim3 grid((unsigned)((total * 8 / 16 + 255) / 256)), block(256);
            if (wnt) hipLaunchKernelGGL((dswx_write_probe_k<0, true>), grid, block, 0, s, a, total);
            else hipLaunchKernelGGL((dswx_write_probe_k<0, false>), grid, block, 0, s, a, total);
        } else if (wm == 1) {
            dim3 grid((unsigned)((total / 16 + 255) / 256), 1, 7), block(256);
            if (wnt) hipLaunchKernelGGL((dswx_write_probe_k<1, true>), grid, block, 0, s, a, total);
            else hipLaunchKernelGGL((dswx_write_probe_k<1, false>), grid, block, 0, s, a, total);
        } else {
            dim3 grid((unsigned)((total + 4095) / 4096)), block(256);
            if (wnt) hipLaunchKernelGGL((dswx_write_probe_k<2, true>), grid, block, 0, s, a, total);
            else hipLaunchKernelGGL((dswx_write_probe_k<2, false>), grid, block, 0, s, a, total);
        }
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    const bool ppt16 = variant & 1, nt = variant & 2;
    const int iters = 1 << ((variant >> 2) & 7);
    const int mode = (variant >> 9) & 3;
    const bool xcd = variant & 2048, big = variant & 4096;
    const int ppt = ppt16 ? 16 : 8, bs = big ? 512 : 256;
    const int64_t groups = n_pixels / ppt;
    dim3 grid((unsigned)((groups + (int64_t)bs * iters - 1) / ((int64_t)bs * iters)), (unsigned)n_tiles), block(bs);
#define PROBE_LAUNCH(PPT, NT, MODE, XCD, BS) hipLaunchKernelGGL((dswx_stream_probe_k<PPT, NT, MODE, XCD, BS>), grid, block, 0, s, a, iters)
#define PROBE_SEL5(PPT, NT, MODE, XCD) do { if (big) PROBE_LAUNCH(PPT, NT, MODE, XCD, 512); else PROBE_LAUNCH(PPT, NT, MODE, XCD, 256); } while (0)
#define PROBE_SEL4(PPT, NT, MODE) do { if (xcd) PROBE_SEL5(PPT, NT, MODE, true); else PROBE_SEL5(PPT, NT, MODE, false); } while (0)
#define PROBE_SEL3(PPT, NT) do { if (mode == 0) PROBE_SEL4(PPT, NT, 0); else if (mode == 1) PROBE_SEL4(PPT, NT, 1); else PROBE_SEL4(PPT, NT, 2); } while (0)
#define PROBE_SEL2(PPT) do { if (nt) PROBE_SEL3(PPT, true); else PROBE_SEL3(PPT, false); } while (0)
    if (ppt16) PROBE_SEL2(16); else PROBE_SEL2(8);
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

int dswx_synth_fill(dswx_ctx_t* ctx, uint64_t seed, int64_t tile0, int64_t n_tiles, int64_t height,
                    int64_t width, const dswx_planes_in_t* in, void* stream) {
    if (!ctx || !in) return fail(DSWX_ERR_ARG, "NULL argument");
    if (n_tiles < 0 || height < 0 || width < 0 || tile0 < 0) return fail(DSWX_ERR_ARG, "negative size");
    for (int k = 0; k < 6; ++k)
        if (!in->band[k]) return fail(DSWX_ERR_ARG, "band[%d] is NULL", k);
    if (!in->fmask) return fail(DSWX_ERR_ARG, "fmask is NULL");
    const int64_t P = height * width;
    if (n_tiles == 0 || P == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int64_t max_y = 65535;
    for (int64_t t0 = 0; t0 < n_tiles; t0 += max_y) {
        const int64_t nt = (n_tiles - t0 < max_y) ? n_tiles - t0 : max_y;
        dswx_planes_in_t b = *in;
        const int64_t shift = t0 * P;
        for (int k = 0; k < 6; ++k) b.band[k] += shift;
        b.fmask += shift;
        if (b.land) b.land += shift;
        if (b.shad) b.shad += shift;
        if (b.ocean) b.ocean += shift;
        dim3 grid((unsigned)((P + 255) / 256), (unsigned)nt), block(256);
        hipLaunchKernelGGL(dswx_synth_v1, grid, block, 0, s, b, (unsigned long long)seed,
                           (long long)(tile0 + t0), (long long)P, (int)width);
        HIP_TRY(hipGetLastError());
    }
    return DSWX_OK;
}

int dswx_device_malloc(dswx_ctx_t* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return fail(DSWX_ERR_ARG, "NULL argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
    return DSWX_OK;
}

int dswx_device_free(dswx_ctx_t* ctx, void* ptr) {
    if (!ctx) return fail(DSWX_ERR_ARG, "NULL argument");
    if (!ptr) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipFree(ptr));
    return DSWX_OK;
}

int dswx_memcpy_h2d(dswx_ctx_t* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return DSWX_OK;
}

int dswx_memcpy_d2h(dswx_ctx_t* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return DSWX_OK;
}

int dswx_memset_d(dswx_ctx_t* ctx, void* dst, int value, size_t bytes) {
    if (!ctx) return fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemset(dst, value, bytes));
    return DSWX_OK;
}

int dswx_stream_synchronize(dswx_ctx_t* ctx, void* stream) {
    if (!ctx) return fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
    return DSWX_OK;
}

int dswx_event_create(dswx_ctx_t* ctx, void** out) {
    if (!ctx || !out) return fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *out = e;
    return DSWX_OK;
}

int dswx_event_destroy(dswx_ctx_t* ctx, void* event) {
    if (!ctx) return fail(DSWX_ERR_ARG, "NULL argument");
    if (event) HIP_TRY(hipEventDestroy((hipEvent_t)event));
    return DSWX_OK;
}

int dswx_event_record(dswx_ctx_t* ctx, void* event, void* stream) {
    if (!ctx || !event) return fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord((hipEvent_t)event, stream ? (hipStream_t)stream : ctx->stream));
    return DSWX_OK;
}

int dswx_event_elapsed_ms(dswx_ctx_t* ctx, void* start, void* stop, float* ms) {
    if (!ctx || !start || !stop || !ms) return fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return DSWX_OK;
}

int dswx_last_kernel_info(dswx_ctx_t* ctx, char* buf, size_t buflen) {
    if (!ctx || !buf || buflen == 0) return fail(DSWX_ERR_ARG, "NULL argument");
    snprintf(buf, buflen, "%s", ctx->last_kernel.c_str());
    return DSWX_OK;
}

}  // extern "C"
