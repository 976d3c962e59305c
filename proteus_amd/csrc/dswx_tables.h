// dswx_tables.h -- the table-driven classification of one 8-pixel group: device tables,
// packed-int16 arithmetic, byte transposes.  Shared by the production kernel
// (dswx_classify_lut.hip) and the experimental data-movement variants (dswx_variants.hip).
#pragma once

#include <cstring>
#include <limits>

#include "dswx_device.h"
// ==============================================================================
// Table-driven per-pixel chain
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
                           //   | is_fill<<8 | prelim_cloud_nonzero<<9 | adjacent(bit 2)<<10
    uint8_t land8[256];    // LAND byte -> is_water(200) | psw_rule_class(201 or <100)<<1 | high_dev<<2
    // without LAND / SHAD planes (WTR-2 = WTR-1-AEROSOL): one lookup
    uint2 chain[128];      // [code | remap<<3 | shadow<<4 | cloud<<5 | snow<<6]
                           //   -> x = WTR-1-AEROSOL | WTR-2<<8 | WTR<<16 | BWTR<<24,
                           //      y = CONF | CLOUD<<8 | valid<<16 | cloud_and_valid<<24
    uint2 extra[256];      // chain index | Fmask adjacent << 7 -> x = cover state byte (cover_state_of | adjacent << 7:
                           //   stage 1 of 'cover' mode) | browse << 16, y = the four dilation predicates of the
                           //   pixel spread over the bytes (cover_spread_of); read only by the EXTRAS kernels
    // with LAND / SHAD planes the chain factors at WTR-2 into two 128-entry lookups (2.5 KiB of
    // tables per block instead of the 9.5 KiB of a flat 1024-entry table, whose per-block load
    // cost 5 % of the kernel):
    uint16_t pre16[128];   // [code | remap<<3 | shadrule<<4 | lcpsw<<5 | lchigh<<6]
                           //   -> WTR-1-AEROSOL as saved | WTR-2 code << 8
    uint2 chainm[128];     // [WTR-2 code | remap<<3 | shadow<<4 | cloud<<5 | snow<<6] -> x, y as `chain`
                           //   (byte 0 of x unused: WTR-1-AEROSOL comes from pre16)
    uint2 extram[256];     // chainm index | adjacent << 7, content as `extra`
};
// WTR-1 / WTR-2 "code": 0..4 = class, 5 = ocean masked (254), 6 = fill (255)

static __global__ __launch_bounds__(256) void dswx_build_tables(const DevParams P, Tables* __restrict__ t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t cc = (uint32_t)P.collapse;
    auto class_of = [](uint32_t code) { return code <= 4u ? code : (code == 5u ? 254u : 255u); };
    auto code_of = [](uint32_t cls) { return cls <= 4u ? cls : (cls == 254u ? 5u : 6u); };
    if (i < 128) {
        const uint32_t dd = (uint32_t)(i & 3) | ((((uint32_t)i >> 2) & 1u) ^ 1u) << 2 | (((uint32_t)i >> 3) & 3u) << 3;
        uint32_t diag, w1;
        px_w1(dd, (i >> 5) & 1, (i >> 6) & 1, diag, w1);
        t->lut1[i] = diag | (code_of(w1) << 16) | (collapse_class(w1, cc) << 24);
        // y bits 16 / 24: the pixel's contributions to n_valid and n_cloud_and_valid (A3)
        const uint32_t code = i & 7, remap = (i >> 3) & 1, shadow = (i >> 4) & 1, cloud = (i >> 5) & 1, snow = (i >> 6) & 1;
        const uint32_t valid = code < 5u ? 1u : 0u, pc_nz = shadow | cloud;
        PxOut o;
        px_chain(P, class_of(code), remap, shadow + 4u * cloud, snow, false, false, false, o);
        t->chain[i] = make_uint2(o.wtr1a | o.wtr2 << 8 | o.wtr << 16 | o.bwtr << 24,
                                 o.conf | o.cloud << 8 | valid << 16 | (valid & pc_nz) << 24);
        // first factor: here bits 4..6 of i are the three LAND / SHAD rule hits
        px_chain(P, class_of(code), remap, 0u, false, (i >> 4) & 1, (i >> 5) & 1, (i >> 6) & 1, o);
        t->pre16[i] = (uint16_t)(o.wtr1a | code_of(o.w2_raw) << 8);
        // second factor: `code` is now the WTR-2 code; the aerosol flag of CLOUD (bit 3) needs the
        // remap bit again (do_remap = remap & class <= 4, and A10 never leaves or enters 0..4)
        const uint32_t w2 = class_of(code), pc = (shadow + 4u * cloud) | ((remap && code < 5u) ? 8u : 0u);
        finish_px(P, w2, pc, snow, o);
        t->chainm[i] = make_uint2(o.wtr2 << 8 | o.wtr << 16 | o.bwtr << 24,
                                  o.conf | o.cloud << 8 | valid << 16 | (valid & pc_nz) << 24);
    }
    if (i < 256) {
        {   // EXTRAS tables: bits 0-6 = the chain index of the pixel, bit 7 = its Fmask adjacent bit
            const uint32_t code = i & 7, remap = (i >> 3) & 1, shadow = (i >> 4) & 1, cloud = (i >> 5) & 1, snow = (i >> 6) & 1;
            const uint32_t adj = ((uint32_t)i >> 7) << 7;
            PxOut o;
            px_chain(P, class_of(code), remap, shadow + 4u * cloud, snow, false, false, false, o);
            t->extra[i] = make_uint2(o.state | adj | o.browse << 16, cover_spread_of(o.state | adj));
            const uint32_t w2 = class_of(code), pc = (shadow + 4u * cloud) | ((remap && code < 5u) ? 8u : 0u);
            finish_px(P, w2, pc, snow, o);
            const uint32_t st = cover_state_of(w2, pc, snow != 0u) | adj;
            t->extram[i] = make_uint2(st | o.browse << 16, cover_spread_of(st));
        }
        const uint32_t aer = (P.aer_lut[i >> 2] >> (8 * (i & 3))) & 0x1fu;
        const uint32_t shadow = (i & P.shadow_bits) ? 1u : 0u, cloud = (i >> 1) & 1u, snow = (i >> 4) & 1u;
        t->fm16[i] = (uint16_t)(aer | shadow << 5 | cloud << 6 | snow << 7 | (i == P.fmask_fill ? 1u : 0u) << 8 |
                                (shadow | cloud) << 9 | (((uint32_t)i >> 2) & 1u) << 10);
        t->land8[i] = (uint8_t)((i == 200 ? 1 : 0) | ((i == 201 || i < 100) ? 2 : 0) | ((i >= 100 && i < 200) ? 4 : 0));
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
// F32 (flag_offset_and_scale_inputs, :2300-2302): the five tests and the two nir comparisons of A9 / A10 on float32
// reflectances scale * (float32(x) - offset), operation by operation as classify_px_f32 (dswx_device.h) and numpy
// evaluate them (-ffp-contract=off, IEEE division); fill test, clip, tables and packing are the integer path's.
template <bool MASKS, bool WANT_IDX = false, bool F32 = false>
__device__ __forceinline__ void lut_group(const DevParams& P, const LutConsts& C, const uint32_t* __restrict__ s_lut1,
                                          const uint16_t* __restrict__ s_fm16, const uint8_t* __restrict__ s_land8,
                                          const uint2* __restrict__ s_chain, const uint16_t* __restrict__ s_pre16,
                                          const u32x4 (&v)[6], const u32x2 vf,
                                          const u32x2 vl, const u32x2 vs, const u32x2 vo, bool has_l, bool in_range,
                                          uint32_t (&w1w)[8], uint32_t (&chx)[8], uint32_t (&chy)[8], uint32_t& cnt,
                                          uint32_t* idx_out = nullptr) {
        uint32_t gsum = 0;
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
            uint32_t d1 = 0, n1 = 0, mn = 0, n2 = 0, d2 = 0, t2s = 0, viol4 = 0, viol5 = 0, dark = 0, noaer = 0;
            if (!F32) {
                d1 = pk_add(g, s1); n1 = pk_sub(g, s1); mn = pk_add(n, s1);
                n2 = pk_sub(n, r); d2 = pk_add(n, r);
                // sign bit (15 / 31) set  <=>  ...
                t2s = pk_sub_sat(mn, pk_add(g, r));                                                 // T2 true
                viol4 = pk_sub_sat(C.k_p1_swir1, s1) | pk_sub_sat(C.k_p1_nir, n) | C.force4;        // T4 ints fail
                viol5 = pk_sub_sat(C.k_p2_blue, b) | pk_sub_sat(C.k_p2_swir1, s1) |
                        pk_sub_sat(C.k_p2_swir2, s2) | pk_sub_sat(C.k_p2_nir, n) | C.force5;        // T5 ints fail
                dark = pk_sub_sat(n, C.k_lc_nir) | C.force_dark;                                    // nir NOT > lcmask_nir
                noaer = pk_sub_sat(C.k_aer_nir, n) | C.force_noaer;                                 // nir NOT <= 1000
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int j = wd * 2 + hf, sh = 16 * hf;
                const int bw = j >> 2, bk = j & 3;
                uint32_t t1, t2, t3n, t4, t5, dark1, noaer1;
                if (F32) {
                    const float fb = P.f_scale[0] * ((float)s16_of(b, hf) - P.f_offset[0]);
                    const float fg = P.f_scale[1] * ((float)s16_of(g, hf) - P.f_offset[1]);
                    const float fr = P.f_scale[2] * ((float)s16_of(r, hf) - P.f_offset[2]);
                    const float fn = P.f_scale[3] * ((float)s16_of(n, hf) - P.f_offset[3]);
                    const float fs1 = P.f_scale[4] * ((float)s16_of(s1, hf) - P.f_offset[4]);
                    const float fs2 = P.f_scale[5] * ((float)s16_of(s2, hf) - P.f_offset[5]);
                    const float mndwi = (fg - fs1) / (fg + fs1);
                    const float mbsrv = fg + fr, mbsrn = fn + fs1;
                    const float awesh = ((fb + 2.5f * fg) - 1.5f * mbsrn) - 0.25f * fs2;
                    const float ndvi = (fn - fr) / (fn + fr);
                    const float* T = P.f_thr;   // wigt awgt p1_mndwi p1_nir p1_swir1 p1_ndvi p2_mndwi p2_blue p2_nir p2_swir1 p2_swir2 lcmask_nir
                    t1 = mndwi > T[0] ? 1u : 0u;
                    t2 = mbsrv > mbsrn ? 1u : 0u;
                    t3n = awesh > T[1] ? 0u : 1u;
                    t4 = ((mndwi > T[2]) & (fs1 < T[4]) & (fn < T[3]) & (ndvi < T[5])) ? 1u : 0u;
                    t5 = ((mndwi > T[6]) & (fb < T[7]) & (fs1 < T[9]) & (fs2 < T[10]) & (fn < T[8])) ? 1u : 0u;
                    dark1 = fn > T[11] ? 0u : 1u;
                    noaer1 = fn <= P.f_aer_nir ? 0u : 1u;
                } else {
                    // ---- A4 quotient tests, sign-bit form (see the header comment)
                    const int in1 = s16_of(n1, hf), id1 = s16_of(d1, hf), in2 = s16_of(n2, hf), id2 = s16_of(d2, hf);
                    const double dn1 = (double)in1, dd1 = (double)id1, dn2 = (double)in2, dd2 = (double)id2;
                    const double r0 = __builtin_fma(-P.qt[0], dd1, dn1), r1 = __builtin_fma(-P.qt[1], dd1, dn1),
                                 r2 = __builtin_fma(-P.qt[2], dd1, dn1), r3 = __builtin_fma(-P.qt[3], dd2, dn2);
                    // sign(h*d - r) = 1  <=>  r > h*d ;  sign(r - hneg*d) = 1  <=>  r < hneg*d ; exact zero -> +0
                    const uint32_t g0 = hi32(__builtin_fma(P.qh[0], dd1, -r0)) >> 31, g1 = hi32(__builtin_fma(P.qh[1], dd1, -r1)) >> 31,
                                   g2 = hi32(__builtin_fma(P.qh[2], dd1, -r2)) >> 31, l3 = hi32(__builtin_fma(-P.qh[3], dd2, r3)) >> 31;
                    const uint32_t neg1 = (uint32_t)id1 >> 31, neg2 = (uint32_t)id2 >> 31;
                    const uint32_t m_p1 = g1 ^ neg1, m_p2 = g2 ^ neg1, v_p1 = l3 ^ neg2;
                    t1 = g0 ^ neg1;
                    // ---- AWESH as int32: sign set <=> 4*awesh < awesh4_min  (T3 false)
                    const int aw = C.awesh_init + 4 * s16_of(b, hf) + 10 * s16_of(g, hf) - 6 * s16_of(mn, hf) - s16_of(s2, hf);
                    t3n = (uint32_t)aw >> 31;
                    t2 = (t2s >> (15 + sh)) & 1u;
                    t4 = m_p1 & v_p1 & ~(viol4 >> (15 + sh)) & 1u;
                    t5 = m_p2 & ~(viol5 >> (15 + sh)) & 1u;
                    dark1 = (dark >> (15 + sh)) & 1u;
                    noaer1 = (noaer >> (15 + sh)) & 1u;
                }
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
                const uint32_t remap = (F >> code) & ~noaer1 & 1u;
                uint32_t idx2 = code | remap << 3 | ((F >> 5) & 7u) << 4;
                uint32_t w1a = 0;
                if (MASKS) {
                    // first factor (LAND / SHAD rules): WTR-1-AEROSOL and the WTR-2 code that indexes the second
                    const uint32_t shadrule = (shad_nz ^ 1u) & ~lbits & 1u;
                    const uint32_t lcpsw = (lbits >> 1) & ~dark1 & 1u;
                    const uint32_t pre = s_pre16[code | remap << 3 | shadrule << 4 | lcpsw << 5 | ((lbits >> 2) & 1u) << 6];
                    w1a = pre & 0xffu;
                    idx2 = (pre >> 8) | remap << 3 | ((F >> 5) & 7u) << 4;
                }
                const uint2 ch = s_chain[idx2];
                w1w[j] = word1; chx[j] = MASKS ? (ch.x | w1a) : ch.x; chy[j] = ch.y;
                if (WANT_IDX) idx_out[j] = idx2 | ((F >> 3) & 128u);       // chain index | adjacent << 7: index of Tables::extra
                gsum += ch.y >> 16;                  // A3: valid | cloud_and_valid << 8, from the table
            }
        }
        cnt += in_range ? (gsum & 0xffu) | (gsum >> 8) << 16 : 0u;
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

// ---- host side: packed / derived constants of one launch
static inline uint32_t pack16(int v) { return ((uint32_t)v & 0xffffu) * 0x10001u; }

static inline void make_lut_consts(const DevParams& d, LutConsts* c) {
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
