// dswx_hip.hip -- MI355X (gfx950 / CDNA4) DSWx-HLS per-pixel classifier: the direct fused kernel,
// the generic kernel, kernel dispatch and the core of the C-ABI
// (context, parameters, device-pointer entry points, device plumbing).  Elsewhere:
//   dswx_device.h         per-pixel device functions (single source of truth of the chain)
//   dswx_classify_lut.hip table-driven production kernel (+ dswx_tables.h)
//   dswx_host_path.hip    dswx_classify_host (synchronous and pipelined), page-locked memory
//   dswx_layers.hip       shadow layer, LAND aggregation, interpret-alone, synthetic tiles
//   (roofline probes and A/B switches: libdswx_lab.so, tools/lab/csrc/)
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
// dswx_make_dev_params.)  `<` is the mirror image with the lower midpoint.  tests/ checks this
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

#include "dswx_host.h"

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
template <bool MASKS, bool EXTRAS, int WPS = 4, bool F32 = false>
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
        const long long off = (long long)blockIdx.y * a.tile_stride + (in_range ? grp : n_groups - 1) * 8;
        u32x4 v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = ldg_u<u32x4_u, u32x4, true>(a.in.band[k] + off);
        const u32x2 vf = ldg_u<u32x2_u, u32x2, true>(a.in.fmask + off);
        u32x2 vl = {0u, 0u}, vs = {0u, 0u}, vo = {0u, 0u};
        bool has_l = false, has_s = false, has_o = false;
        if (MASKS) {
            has_l = a.in.land != nullptr; has_s = a.in.shad != nullptr; has_o = a.in.ocean != nullptr;
            if (has_l) vl = ldg_u<u32x2_u, u32x2, true>(a.in.land + off);
            if (has_s) vs = ldg_u<u32x2_u, u32x2, true>(a.in.shad + off);
            if (has_o) {
                vo = ldg_u<u32x2_u, u32x2, true>(a.in.ocean + off);
                t_ocean = __builtin_amdgcn_sad_u8(vo.x, 0u, __builtin_amdgcn_sad_u8(vo.y, 0u, 0u));
                t_ocean = in_range ? t_ocean : 0u;
            }
        }
        uint32_t q_diag[4] = {0, 0, 0, 0};
        uint32_t cflags[EXTRAS ? 8 : 1] = {};
        uint32_t q_w1[2] = {0, 0}, q_w1a[2] = {0, 0}, q_w2[2] = {0, 0}, q_w[2] = {0, 0},
                 q_bw[2] = {0, 0}, q_cf[2] = {0, 0}, q_cl[2] = {0, 0}, q_st[2] = {0, 0}, q_br[2] = {0, 0};
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
            if (F32) classify_px_f32(P, aer_bits, b, g, r, n, s1, s2, fm, land, shad, ocean, o, ok, cv);
            else classify_px(P, aer_bits, b, g, r, n, s1, s2, fm, land, shad, ocean, o, ok, cv);
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
                q_st[bw] |= (o.state | ((uint32_t)fm & 4u) << 5) << (8 * bk);
                cflags[j] = cover_flags_of(o.state | ((uint32_t)fm & 4u) << 5);
                q_br[bw] |= o.browse << (8 * bk);
            }
        }
        if (in_range) {
        if (a.out.diag) stg_u<u32x4_u, u32x4, true>(a.out.diag + off, u32x4{q_diag[0], q_diag[1], q_diag[2], q_diag[3]});
        if (a.out.wtr1) stg_u<u32x2_u, u32x2, true>(a.out.wtr1 + off, u32x2{q_w1[0], q_w1[1]});
        if (a.out.wtr1_aerosol) stg_u<u32x2_u, u32x2, true>(a.out.wtr1_aerosol + off, u32x2{q_w1a[0], q_w1a[1]});
        if (a.out.wtr2) stg_u<u32x2_u, u32x2, true>(a.out.wtr2 + off, u32x2{q_w2[0], q_w2[1]});
        if (a.out.wtr) stg_u<u32x2_u, u32x2, true>(a.out.wtr + off, u32x2{q_w[0], q_w[1]});
        if (a.out.bwtr) stg_u<u32x2_u, u32x2, true>(a.out.bwtr + off, u32x2{q_bw[0], q_bw[1]});
        if (a.out.conf) stg_u<u32x2_u, u32x2, true>(a.out.conf + off, u32x2{q_cf[0], q_cf[1]});
        if (a.out.cloud) stg_u<u32x2_u, u32x2, true>(a.out.cloud + off, u32x2{q_cl[0], q_cl[1]});
        if (EXTRAS && a.out.browse) stg_u<u32x2_u, u32x2, true>(a.out.browse + off, u32x2{q_br[0], q_br[1]});
        if (EXTRAS && a.cover_state) {   // 'cover' stage 1 (wave-uniform)
            *reinterpret_cast<u32x2_u*>(a.cover_state + off) = u32x2{q_st[0], q_st[1]};
            uint32_t fl[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) fl[j] = cflags[EXTRAS ? j : 0];
            a.cover_bits[(long long)blockIdx.y * a.cover_bits_stride + grp] = cover_bits_of(fl);
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

// Sums the fused kernel's per-wave partial counts of one tile (block = tile) and
// WRITES counters[tile]; the ragged-remainder kernel adds to them afterwards.
__global__ __launch_bounds__(1024) void dswx_counters_finish(const uint2* __restrict__ partials,
                                                             unsigned long long* __restrict__ counters,
                                                             long long per_tile, int has_ocean,
                                                             long long vec_pixels, const uint8_t* ragged_fmask,
                                                             long long tile_stride, long long n_pixels) {
    // one block of 1024 threads per tile, four independent loads in flight per thread: the
    // 26 k partials of a 3660 x 3660 tile are summed in ~7 dependent rounds (this kernel is pure
    // latency; with 256 threads and one load at a time it took 30 us, a third of a single-tile call)
    __shared__ unsigned long long red[16][3];
    const uint2* p = partials + (long long)blockIdx.x * per_tile;
    unsigned long long v = 0, c = 0, o = 0;
    long long i = threadIdx.x;
    for (; i + 3 * 1024 < per_tile; i += 4 * 1024) {
        const uint2 x0 = p[i], x1 = p[i + 1024], x2 = p[i + 2048], x3 = p[i + 3072];
        v += (x0.x & 0xffffu) + (x1.x & 0xffffu) + (x2.x & 0xffffu) + (x3.x & 0xffffu);
        c += (x0.x >> 16) + (x1.x >> 16) + (x2.x >> 16) + (x3.x >> 16);
        o += (unsigned long long)x0.y + x1.y + x2.y + x3.y;
    }
    for (; i < per_tile; i += 1024) {
        const uint2 x = p[i];
        v += x.x & 0xffffu; c += x.x >> 16; o += x.y;
    }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
        v += __shfl_xor(v, sh); c += __shfl_xor(c, sh); o += __shfl_xor(o, sh);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = v; red[threadIdx.x >> 6][1] = c; red[threadIdx.x >> 6][2] = o; }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long sum = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) sum += red[w][threadIdx.x];
        if (threadIdx.x == 2 && !has_ocean) {
            if (ragged_fmask) {         // KArgs::ragged: the pixels the vector kernel covered differ from tile to tile
                const int head = ragged_head(ragged_fmask + (long long)blockIdx.x * tile_stride);
                vec_pixels = n_pixels > head ? ((n_pixels - head) >> 3) * 8 : 0;
            }
            sum = (unsigned long long)vec_pixels;
        }
        counters[(long long)blockIdx.x * 3 + threadIdx.x] = sum;
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
    uint32_t c0 = 0, c1 = 0, c2 = 0, cstate = 0, cflags = 0;
    long long px = a.px_begin + (long long)blockIdx.x * 256 + threadIdx.x;
    bool active = px < a.n_pixels;
    if (a.ragged) {
        // the edges of a ragged tile behind the table-driven kernel (KArgs::ragged; grid.x = 1): threads 0-7 its head
        // pixels, threads 8-15 its tail pixels
        const int head = ragged_head(a.in.fmask + (long long)blockIdx.y * a.tile_stride);
        const long long body = a.n_pixels > head ? ((a.n_pixels - head) >> 3) * 8 : 0;
        const int t = threadIdx.x;
        px = t < 8 ? t : head + body + (t - 8);
        active = t < 16 && px < a.n_pixels && (t < 8 ? t < head : true);
    }
    if (active) {
        const long long off = (long long)blockIdx.y * a.tile_stride + px;
        int land = -1, shad = 1, ocean = 1;
        if (a.in.land) land = a.in.land[off];
        if (a.in.shad) shad = a.in.shad[off];
        if (a.in.ocean) ocean = a.in.ocean[off];
        PxOut o;
        bool ok, cv;
        const int fm = a.in.fmask[off];
        if (a.P.f32_mode)       // flag_offset_and_scale_inputs: the chain on float32 reflectances (block-uniform)
            classify_px_f32(a.P, lut[fm], a.in.band[0][off], a.in.band[1][off], a.in.band[2][off], a.in.band[3][off],
                            a.in.band[4][off], a.in.band[5][off], fm, land, shad, ocean, o, ok, cv);
        else
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
        if (a.cover_state) {
            cstate = o.state | ((uint32_t)fm & 4u) << 5;
            a.cover_state[off] = (uint8_t)cstate;
            cflags = cover_flags_of(cstate);
        }
    }
    if (a.cover_state) {
        // the wave's 64 consecutive pixels start on a multiple of 8 (px_begin is one): four ballots are the
        // bitmaps of its eight 8-pixel groups; lane k stores group k's dword (pixels past the tile's end: 0)
        unsigned long long bal[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bal[q] = __ballot((cflags >> q) & 1u);
        const int lane = threadIdx.x & 63;
        const long long g = ((px - lane) >> 3) + lane;
        if (lane < 8 && (px - lane) + 8 * lane < a.n_pixels) {
            uint32_t m = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) m |= (uint32_t)((bal[q] >> (8 * lane)) & 0xffull) << (8 * q);
            a.cover_bits[(long long)blockIdx.y * a.cover_bits_stride + g] = m;
        }
    }
    if (a.counters) reduce_counters(a.counters + (long long)blockIdx.y * 3, red, c0, c1, c2);
}

// ------------------------------------------------------------------------------
// Debug planes: float64 MNDWI / NDVI / AWESH exactly as :1872-1887 (true IEEE
// division; int16 wrap-around sums).  Not on the timed path.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dswx_indices_v1(const KArgs a) {
    const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
    if (px >= a.n_pixels) return;
    const long long off = (long long)blockIdx.y * a.tile_stride + px;
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

// ==============================================================================
// host side
// ==============================================================================

static thread_local std::string g_err;

int dswx_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}


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

int dswx_make_dev_params(const dswx_params_t* p, DevParams* d) {
    const double thr[12] = {p->wigt, p->awgt, p->pswt_1_mndwi, p->pswt_1_nir, p->pswt_1_swir1,
                            p->pswt_1_ndvi, p->pswt_2_mndwi, p->pswt_2_blue, p->pswt_2_nir,
                            p->pswt_2_swir1, p->pswt_2_swir2, p->lcmask_nir};
    for (double t : thr)
        if (!std::isfinite(t) || std::fabs(t) > 1e100 || (t != 0.0 && std::fabs(t) < 1e-290))
            return dswx_fail(DSWX_ERR_ARG, "HLS thresholds must be finite, |t| <= 1e100, and 0 or |t| >= 1e-290");
    if (!std::isfinite(p->aerosol_max_nir)) return dswx_fail(DSWX_ERR_ARG, "aerosol_max_nir must be finite");
    if (p->mask_adjacent_to_cloud_mode < 0 || p->mask_adjacent_to_cloud_mode > 2)
        return dswx_fail(DSWX_ERR_UNSUPPORTED, "ERROR mask adjacent to cloud/cloud-shadow mode: %d",
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
    d->f32_mode = p->offset_and_scale_inputs ? 1 : 0;
    if (d->f32_mode) {
        for (int i = 0; i < 6; ++i) {
            if (!std::isfinite(p->band_scale[i]) || !std::isfinite(p->band_offset[i]))
                return dswx_fail(DSWX_ERR_ARG, "band_scale / band_offset must be finite when offset_and_scale_inputs is set");
            d->f_scale[i] = (float)p->band_scale[i];
            d->f_offset[i] = (float)p->band_offset[i];
        }
        for (int i = 0; i < 12; ++i) d->f_thr[i] = (float)thr[i];
        d->f_aer_nir = (float)p->aerosol_max_nir;
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


std::mutex& dswx_va_mutex() { static std::mutex* m = new std::mutex; return *m; }      // never destroyed: frees may arrive during exit

extern "C" {

int dswx_abi_version(void) { return DSWX_ABI_VERSION; }

const char* dswx_last_error(void) { return g_err.c_str(); }

int dswx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dswx_params_default(dswx_params_t* p) {
    if (!p) return dswx_fail(DSWX_ERR_ARG, "params is NULL");
    std::memset(p, 0, sizeof *p);
    p->wigt = 0.124; p->awgt = 0.0;
    p->pswt_1_mndwi = -0.44; p->pswt_1_nir = 1500; p->pswt_1_swir1 = 900; p->pswt_1_ndvi = 0.7;
    p->pswt_2_mndwi = -0.5; p->pswt_2_blue = 1000; p->pswt_2_nir = 2500; p->pswt_2_swir1 = 3000;
    p->pswt_2_swir2 = 1000; p->lcmask_nir = 1200;
    for (int i = 0; i < 6; ++i) { p->band_fill[i] = -9999.0; p->band_scale[i] = 1.0; }      // scale 1, offset 0: only read when offset_and_scale_inputs is set
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
    if (!out) return dswx_fail(DSWX_ERR_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return dswx_fail(DSWX_ERR_NO_DEVICE, "no HIP device visible: the DSWx HIP path has no CPU fallback");
    if (device < 0 || device >= n) return dswx_fail(DSWX_ERR_ARG, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    dswx_ctx* c = new dswx_ctx();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return dswx_fail(DSWX_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return DSWX_OK;
}

int dswx_ctx_destroy(dswx_ctx_t* ctx) {
    if (!ctx) return DSWX_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->fold_acc) (void)hipFree(ctx->fold_acc);
    if (ctx->cover) (void)hipFree(ctx->cover);
    if (ctx->untile_tmp) (void)hipFree(ctx->untile_tmp);
    if (ctx->tables) (void)hipFree(ctx->tables);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->h2d_stream) (void)hipStreamDestroy(ctx->h2d_stream);
    if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
    for (int i = 0; i < 3; ++i) {
        if (ctx->pipe_in[i]) (void)hipEventDestroy(ctx->pipe_in[i]);
        if (ctx->pipe_k[i]) (void)hipEventDestroy(ctx->pipe_k[i]);
        if (ctx->pipe_out[i]) (void)hipEventDestroy(ctx->pipe_out[i]);
    }
    if (ctx->pipe_counters) (void)hipHostFree(ctx->pipe_counters);
    if (ctx->ws_event) (void)hipEventDestroy(ctx->ws_event);
    delete ctx;
    return DSWX_OK;
}


// One launch of a context at a time on the GPU (dswx_host.h: the workspaces are shared): a call on another stream than
// the previous call's waits for it.  The steady state -- every call on one stream -- costs nothing.
static int dswx_ws_enter(dswx_ctx* ctx, hipStream_t s) {
    if (ctx->ws_used && ctx->ws_last != s) {
        if (!ctx->ws_event) HIP_TRY(hipEventCreateWithFlags(&ctx->ws_event, hipEventDisableTiming));
        // the context's own stream is alive as long as the context: mark its work now; a caller's stream may be gone by
        // now, so its mark was set right behind its launch (dswx_ws_leave)
        if (ctx->ws_last == ctx->stream) HIP_TRY(hipEventRecord(ctx->ws_event, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(s, ctx->ws_event, 0));
    }
    return DSWX_OK;
}
static int dswx_ws_leave(dswx_ctx* ctx, hipStream_t s) {
    if (s != ctx->stream) {
        if (!ctx->ws_event) HIP_TRY(hipEventCreateWithFlags(&ctx->ws_event, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ctx->ws_event, s));
    }
    ctx->ws_last = s;
    ctx->ws_used = true;
    return DSWX_OK;
}

// height/width are only needed (and only trusted) in 'cover' mode; 0 = unknown
static int classify_device_impl(dswx_ctx_t* ctx, const dswx_params_t* params, int64_t n_tiles, int64_t n_pixels,
                                int64_t height, int64_t width, int64_t tile_stride, const dswx_planes_in_t* in,
                                const dswx_planes_out_t* out, int64_t* counters, void* stream) {
    if (!ctx || !params || !in || !out) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (n_tiles < 0 || n_pixels < 0) return dswx_fail(DSWX_ERR_ARG, "negative size");
    if (tile_stride == 0) tile_stride = n_pixels;
    if (tile_stride < n_pixels) return dswx_fail(DSWX_ERR_ARG, "tile_stride smaller than the tile");
    for (int k = 0; k < 6; ++k)
        if (!in->band[k]) return dswx_fail(DSWX_ERR_ARG, "band[%d] is NULL", k);
    if (!in->fmask) return dswx_fail(DSWX_ERR_ARG, "fmask is NULL");
    KArgs a = {};
    int rc = dswx_make_dev_params(params, &a.P);
    if (rc) return rc;
    const bool cover = params->mask_adjacent_to_cloud_mode == DSWX_ADJ_COVER;
    if (cover && (height <= 0 || width <= 0 || height * width != n_pixels))
        return dswx_fail(DSWX_ERR_UNSUPPORTED,
                    "mask_adjacent_to_cloud_mode 'cover' is a 2-D neighbourhood operation: use "
                    "dswx_classify_device_2d / dswx_classify_host, which know the tile height and width");
    if (cover && (height > 2147483647LL || width > 2147483647LL)) return dswx_fail(DSWX_ERR_ARG, "tile too large");
    for (int k = 0; k < 6; ++k)
        if (!aligned_to(in->band[k], 2)) return dswx_fail(DSWX_ERR_ALIGN, "band[%d] not 2-byte aligned", k);
    if (out->diag && !aligned_to(out->diag, 2)) return dswx_fail(DSWX_ERR_ALIGN, "diag not 2-byte aligned");
    if (counters && !aligned_to(counters, 8)) return dswx_fail(DSWX_ERR_ALIGN, "counters not 8-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (n_tiles == 0 || n_pixels == 0) {
        if (counters && n_tiles > 0) HIP_TRY(hipMemsetAsync(counters, 0, (size_t)n_tiles * 3 * sizeof(int64_t), s));
        ctx->last_kernel = "none (empty input)";
        return DSWX_OK;
    }
    if (int wrc = dswx_ws_enter(ctx, s)) return wrc;
    a.in = *in;
    a.out = *out;
    a.counters = reinterpret_cast<unsigned long long*>(counters);
    a.n_pixels = n_pixels;
    a.tile_stride = tile_stride;
    a.cover_state = nullptr; a.cover_bits = nullptr; a.cover_bits_stride = 0;
    a.cover_snow = nullptr; a.cover_snow_stride = 0;
    dswx_planes_out_t final_out = *out;     // what stages 2 + 3 of 'cover' write
    a.height = (int)height; a.width = (int)width;
    if (cover) {
        // scratch: one state byte per pixel (>= 256 bytes, it doubles as the slack IN FRONT of the bitmaps) +
        // one bitmap dword per 8-pixel group + 64 dwords of slack behind them (the unconditional 16-byte row
        // loads of stage 2) + the final snow bit plane
        const size_t groups_per_tile = ((size_t)n_pixels + 7) / 8;
        const size_t snow_dw_per_tile = ((size_t)n_pixels + 31) / 32 + 1;
        const size_t state_bytes = (((size_t)n_tiles * (size_t)tile_stride + 255) & ~(size_t)255) + 256;
        const size_t bits_bytes = ((size_t)n_tiles * groups_per_tile + 64) * 4;
        const size_t need = state_bytes + bits_bytes + (size_t)n_tiles * snow_dw_per_tile * 4;
        if (need > ctx->cover_bytes) {
            HIP_TRY(hipStreamSynchronize(s));
            if (ctx->cover) HIP_TRY(hipFree(ctx->cover));
            ctx->cover = nullptr; ctx->cover_bytes = 0;
            HIP_TRY(dswx_locked_malloc(&ctx->cover, need));
            ctx->cover_bytes = need;
        }
        a.cover_state = static_cast<uint8_t*>(ctx->cover);
        a.cover_bits = reinterpret_cast<uint32_t*>(a.cover_state + state_bytes);
        a.cover_bits_stride = (long long)groups_per_tile;
        a.cover_snow = reinterpret_cast<uint32_t*>(a.cover_state + state_bytes + bits_bytes);
        a.cover_snow_stride = (long long)snow_dw_per_tile;
        // stage 1 stops before the snow step: these layers come from stages 2 + 3
        a.out.wtr = a.out.bwtr = a.out.conf = a.out.cloud = a.out.browse = nullptr;
    }

    const bool any_index = out->mndwi || out->ndvi || out->awesh;
    if (any_index && a.P.f32_mode)
        return dswx_fail(DSWX_ERR_UNSUPPORTED, "the float64 index planes describe the integer chain; not available with offset_and_scale_inputs");
    const bool masks = in->land || in->shad || in->ocean;
    // Since round 6 BOTH vector kernels take ANY plane address and ANY tile stride: their 16- / 8-byte accesses go through
    // under-aligned vector types (dswx_device.h: u32x4_u, u32x2_u), for which the compiler emits the very same
    // global_load_dwordx4 / dwordx2 -- gfx950 performs unaligned global accesses in hardware, at a cost of 0 - 6 %
    // (profiles/r06_fallback_rates.json) -- and the generated code of the table-driven kernel is identical instruction for
    // instruction to the round-5 build.  So the table-driven kernel is THE kernel for every layout (int16 planes 2-byte
    // aligned: checked above): tile-relative 8-pixel groups from pixel 0 of every tile, its per-tile lead-in from the
    // Fmask plane's address, the generic kernel for the n_pixels % 8 tail.  Up to round 5 planes without 16-byte
    // alignment -- and ragged batches in 'cover' mode -- ran whole tiles on the 1-pixel-per-thread kernel at 0.17 of the
    // HBM peak (profiles/r05_generic_kernel_stats.csv); the direct kernel (dswx_classify_v8) stays behind the lab switch
    // fused_variant = 0 (A/B partner and variant parity tests).
    const bool stride8 = (tile_stride % 8 == 0) || n_tiles == 1;
    uint8_t* const u8outs[] = {out->wtr1, out->wtr1_aerosol, out->wtr2, out->wtr, out->bwtr, out->conf, out->cloud,
                               out->browse};
    // Do all planes START on 256-byte boundaries?  Then every tile of every plane has the same residue modulo 256
    // pixels and the table-driven kernel can put every wave access on a line boundary: directly when the tile stride
    // is a multiple of 256 pixels (the padded batch layout), through its per-tile lead-in otherwise (contiguous
    // [n_tiles][H * W] arrays, the reference's natural layout: 3660 x 3660 = 144 mod 256) -- dswx_classify_lut.hip.
    bool bases256 = true;
    for (int k = 0; k < 6 && bases256; ++k) bases256 = aligned_to(in->band[k], 256);
    bases256 = bases256 && aligned_to(in->fmask, 256) && (!in->land || aligned_to(in->land, 256)) &&
               (!in->shad || aligned_to(in->shad, 256)) && (!in->ocean || aligned_to(in->ocean, 256)) &&
               (!out->diag || aligned_to(out->diag, 256));
    for (uint8_t* p : u8outs) bases256 = bases256 && (!p || aligned_to(p, 256));
    const bool stride256 = (tile_stride % 256 == 0) || n_tiles == 1;
    // ragged contiguous batches (H * W not a multiple of 8, several tiles): the table-driven kernel starts every tile at
    // its first 8-pixel boundary, the generic kernel does the < 8 + < 8 pixels at its edges (KArgs::ragged) -- with
    // 256-byte aligned bases that keeps every access of every tile aligned.  Not in 'cover' mode, whose bitmaps are indexed
    // by tile-relative 8-pixel groups (there, and for bases that are not aligned anyway, every tile starts at its pixel 0
    // and the accesses are unaligned); not when the direct kernel is forced.
    const bool ragged = !stride8 && bases256 && !cover && ctx->fused_variant != 0;
    const bool lut_ok = true;
    // the lead-in is a property of the addresses (correct for any of them); 0 only when every tile start is aligned
    const int lead_max = (bases256 && stride256) ? 0 : 31;  // groups: dswx_lut_geometry

    const int64_t max_y = 65535;
    char info[256];
    for (int64_t t0 = 0; t0 < n_tiles; t0 += max_y) {
        const int64_t nt = (n_tiles - t0 < max_y) ? n_tiles - t0 : max_y;
        KArgs b = a;
        const int64_t shift = t0 * tile_stride;
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
        if (b.cover_state) { b.cover_state += shift; b.cover_bits += t0 * b.cover_bits_stride; b.cover_snow += t0 * b.cover_snow_stride; }
        b.px_begin = 0;
        b.partials = nullptr;
        b.fold_acc = nullptr;
        b.fold_group_log2 = 0;
        const int64_t groups = n_pixels >> 3;                      // (ragged: the most a tile can have)
        b.ragged = (ragged && groups > 0) ? 1 : 0;
        // the finishing kernel of the vector path WRITES the counters; only the generic kernel
        // alone (atomic adds) needs them zeroed first
        if (groups == 0 && b.counters)
            HIP_TRY(hipMemsetAsync(b.counters, 0, (size_t)nt * 3 * sizeof(int64_t), s));
        if (groups > 0) {
            // 'cover' stage 1 and the browse plane: the direct kernel or the table-driven one (3), not
            // the experimental structures
            const bool plain_outputs = !cover && !b.out.browse;
            // automatic choice: the table-driven kernel, whatever the addresses (its per-tile lead-in takes care of
            // strides that are not multiples of 256 pixels and of plane bases that are not 256-byte aligned)
            int vsel = ctx->fused_variant;
            if (vsel != 0 && vsel != 3) vsel = lut_ok ? 3 : 0;
            const bool variant = vsel == 3;
            int threads = 256;
            long long gx_ll = (groups + 255) / 256;
            if (variant) dswx_lut_geometry(ctx, groups, !plain_outputs, lead_max, &threads, &gx_ll);
            const int64_t gx = gx_ll;
            const int waves = threads / 64;
            dim3 grid((unsigned)gx, (unsigned)nt), block(threads);
            // few tiles: the table-driven kernel sums the counters itself (dswx_host.h: DSWX_FOLD_MAX_TILES)
            const bool fold = variant && b.counters && nt <= DSWX_FOLD_MAX_TILES && n_pixels < (1LL << 24) && gx < 65536 &&
                              ctx->tune_fold != 0;
            if (fold) {
                // [tile][1 + groups of 64 blocks] accumulators, one 128-byte line each: zeroed when (re)allocated or after a
                // failed launch, left zero by every launch that completes
                const int gl = dswx_lut_fold_group_log2(!plain_outputs);
                const size_t need = (size_t)nt * (size_t)(1 + ((gx + (1LL << gl) - 1) >> gl)) * 128;
                if (need > ctx->fold_bytes) {
                    HIP_TRY(hipStreamSynchronize(s));
                    if (ctx->fold_acc) HIP_TRY(hipFree(ctx->fold_acc));
                    ctx->fold_acc = nullptr; ctx->fold_bytes = 0;
                    HIP_TRY(dswx_locked_malloc(&ctx->fold_acc, need));
                    ctx->fold_bytes = need;
                    ctx->fold_clean = false;
                }
                if (!ctx->fold_clean) {
                    // (rare: first use, growth, or after a failed launch.  Synchronous, so that a later call on ANOTHER
                    // stream of the caller's cannot overtake the zeroing)
                    HIP_TRY(hipMemsetAsync(ctx->fold_acc, 0, ctx->fold_bytes, s));
                    HIP_TRY(hipStreamSynchronize(s));
                    ctx->fold_clean = true;
                }
                b.fold_acc = ctx->fold_acc;
                b.fold_group_log2 = gl;
            } else if (b.counters) {
                const size_t need = (size_t)nt * (size_t)gx * waves * sizeof(uint2);
                if (need > ctx->partials_bytes) {
                    HIP_TRY(hipStreamSynchronize(s));
                    if (ctx->partials) HIP_TRY(hipFree(ctx->partials));
                    ctx->partials = nullptr; ctx->partials_bytes = 0;
                    HIP_TRY(dswx_locked_malloc(&ctx->partials, need));
                    ctx->partials_bytes = need;
                }
                b.partials = static_cast<uint2*>(ctx->partials);
            }
            if (variant) {
                const int vrc = dswx_lut_launch(ctx, b, masks, grid, block, s, info, sizeof info);
                if (vrc) return vrc;
            } else {
                const bool extras = b.out.browse || b.cover_state;
                if (b.P.f32_mode) {         // flag_offset_and_scale_inputs
                    if (masks && extras) hipLaunchKernelGGL((dswx_classify_v8<true, true, 4, true>), grid, block, 0, s, b);
                    else if (masks) hipLaunchKernelGGL((dswx_classify_v8<true, false, 4, true>), grid, block, 0, s, b);
                    else if (extras) hipLaunchKernelGGL((dswx_classify_v8<false, true, 4, true>), grid, block, 0, s, b);
                    else hipLaunchKernelGGL((dswx_classify_v8<false, false, 4, true>), grid, block, 0, s, b);
                }
                else if (masks && extras) hipLaunchKernelGGL((dswx_classify_v8<true, true>), grid, block, 0, s, b);
                else if (masks) hipLaunchKernelGGL((dswx_classify_v8<true, false>), grid, block, 0, s, b);
                else if (extras) hipLaunchKernelGGL((dswx_classify_v8<false, true>), grid, block, 0, s, b);
                else if (ctx->tune_wps == 6) hipLaunchKernelGGL((dswx_classify_v8<false, false, 6>), grid, block, 0, s, b);
                else if (ctx->tune_wps == 8) hipLaunchKernelGGL((dswx_classify_v8<false, false, 8>), grid, block, 0, s, b);
                else hipLaunchKernelGGL((dswx_classify_v8<false, false, 4>), grid, block, 0, s, b);
                snprintf(info, sizeof info, "dswx_classify_v8<%s,%s%s> (fused, direct stores) grid=(%lld,%lld) block=256",
                         masks ? "true" : "false", extras ? "true" : "false", b.P.f32_mode ? ",f32" : "", (long long)gx,
                         (long long)nt);
            }
            if (hipError_t le = hipGetLastError(); le != hipSuccess) {
                ctx->fold_clean = false;
                return dswx_fail(DSWX_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(le));
            }
            if (b.counters && !fold) {
                hipLaunchKernelGGL(dswx_counters_finish, dim3((unsigned)nt), dim3(1024), 0, s, b.partials,
                                   b.counters, (long long)gx * waves, in->ocean ? 1 : 0, (long long)groups * 8,
                                   b.ragged ? b.in.fmask : nullptr, (long long)tile_stride, (long long)n_pixels);
                HIP_TRY(hipGetLastError());
            }
            b.px_begin = groups * 8;
        }
        if (b.ragged) {                     // head and tail pixels of every tile: 16 threads of one block per tile
            hipLaunchKernelGGL(dswx_classify_v1, dim3(1, (unsigned)nt), dim3(256), 0, s, b);
            const size_t len = strlen(info);
            snprintf(info + len, sizeof info - len, " ragged tiles: edges by dswx_classify_v1");
        } else if (b.px_begin < n_pixels) {
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
            const size_t len = strlen(info);
            const int crc = dswx_cover_stage2_launch(ctx, c2, nt, s, info + len, sizeof info - len);
            if (crc) return crc;
        }
        if (any_index) {
            dim3 grid((unsigned)((n_pixels + 255) / 256), (unsigned)nt), block(256);
            hipLaunchKernelGGL(dswx_indices_v1, grid, block, 0, s, b);
            HIP_TRY(hipGetLastError());
        }
    }
    ctx->last_kernel = info;
    return dswx_ws_leave(ctx, s);
}

int dswx_classify_device(dswx_ctx_t* ctx, const dswx_params_t* params, int64_t n_tiles, int64_t n_pixels,
                         const dswx_planes_in_t* in, const dswx_planes_out_t* out, int64_t* counters,
                         void* stream) {
    return classify_device_impl(ctx, params, n_tiles, n_pixels, 0, 0, 0, in, out, counters, stream);
}

int dswx_classify_device_2d(dswx_ctx_t* ctx, const dswx_params_t* params, int64_t n_tiles, int64_t height,
                            int64_t width, const dswx_planes_in_t* in, const dswx_planes_out_t* out,
                            int64_t* counters, void* stream) {
    if (height < 0 || width < 0) return dswx_fail(DSWX_ERR_ARG, "negative size");
    return classify_device_impl(ctx, params, n_tiles, height * width, height, width, 0, in, out, counters, stream);
}

int dswx_classify_batch(dswx_ctx_t* ctx, const dswx_params_t* params, const dswx_batch_geom_t* geom,
                        const dswx_planes_in_t* in, const dswx_planes_out_t* out, int64_t* counters,
                        void* stream) {
    if (!geom) return dswx_fail(DSWX_ERR_ARG, "geom is NULL");
    if (geom->height < 0 || geom->width < 0 || geom->tile_stride < 0) return dswx_fail(DSWX_ERR_ARG, "negative size");
    return classify_device_impl(ctx, params, geom->n_tiles, geom->height * geom->width, geom->height, geom->width,
                                geom->tile_stride, in, out, counters, stream);
}

int dswx_device_malloc(dswx_ctx_t* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(dswx_locked_malloc(out, bytes ? bytes : 1));
    return DSWX_OK;
}

int dswx_device_free(dswx_ctx_t* ctx, void* ptr) {
    if (!ctx) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (!ptr) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipFree(ptr));
    return DSWX_OK;
}

int dswx_memcpy_h2d(dswx_ctx_t* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return DSWX_OK;
}

int dswx_memcpy_d2h(dswx_ctx_t* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return DSWX_OK;
}

int dswx_memset_d(dswx_ctx_t* ctx, void* dst, int value, size_t bytes) {
    if (!ctx) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemset(dst, value, bytes));
    // hipMemset on device memory returns before the fill has run, and the fill is queued on the NULL stream, which the
    // context's (non-blocking) stream does not wait for: a launch issued right after could be overtaken by it (found by the
    // writer kernels' fuzz test, round 6).  Like the two memcpy entries beside it, this one is complete when it returns.
    HIP_TRY(hipStreamSynchronize(nullptr));
    return DSWX_OK;
}

int dswx_stream_synchronize(dswx_ctx_t* ctx, void* stream) {
    if (!ctx) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
    return DSWX_OK;
}

int dswx_event_create(dswx_ctx_t* ctx, void** out) {
    if (!ctx || !out) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *out = e;
    return DSWX_OK;
}

int dswx_event_destroy(dswx_ctx_t* ctx, void* event) {
    if (!ctx) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (event) HIP_TRY(hipEventDestroy((hipEvent_t)event));
    return DSWX_OK;
}

int dswx_event_record(dswx_ctx_t* ctx, void* event, void* stream) {
    if (!ctx || !event) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord((hipEvent_t)event, stream ? (hipStream_t)stream : ctx->stream));
    return DSWX_OK;
}

int dswx_event_elapsed_ms(dswx_ctx_t* ctx, void* start, void* stop, float* ms) {
    if (!ctx || !start || !stop || !ms) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return DSWX_OK;
}

int dswx_last_kernel_info(dswx_ctx_t* ctx, char* buf, size_t buflen) {
    if (!ctx || !buf || buflen == 0) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    snprintf(buf, buflen, "%s", ctx->last_kernel.c_str());
    return DSWX_OK;
}

}  // extern "C"
