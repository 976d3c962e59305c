// dswx_hip.hip -- MI355X (gfx950 / CDNA4) DSWx-HLS per-pixel classifier: the direct fused kernel,
// the 'cover' stage-2 kernels, the generic kernel, kernel dispatch and the core of the C-ABI
// (context, parameters, device-pointer entry points, device plumbing).  Elsewhere:
//   dswx_device.h         per-pixel device functions (single source of truth of the chain)
//   dswx_classify_lut.hip table-driven production kernel (+ dswx_tables.h)
//   dswx_host_path.hip    dswx_classify_host (synchronous and pipelined), page-locked memory
//   dswx_layers.hip       shadow layer, LAND aggregation, interpret-alone, synthetic tiles
//   dswx_variants.hip     experimental data-movement structures;  dswx_probes.hip  roofline probes
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
#include "dswx_tables.h"     // transpose4 (byte transposes) for the quad cover kernel

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
        const long long off = (long long)blockIdx.y * a.tile_stride + (in_range ? grp : n_groups - 1) * 8;
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
    const long long tile_base = (long long)blockIdx.z * a.tile_stride;
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

// ------------------------------------------------------------------------------
// 'cover' mode, stage 2, bit-packed (default).  Same semantics as dswx_cover_stage2 above, but
// the four per-pixel predicates live as BITMAPS: one block owns a 128-column x 256-row window
// (halo 17 on every side -> 94 x 222 output pixels), one thread owns one window row as a
// 128-bit word pair per mask.  A masked 4-neighbour dilation step is then
//     x |= (x | x<<1 | x>>1 | row_above | row_below) & mask
// on 128 bits: ~30 VALU and one 16-byte LDS exchange per thread and iteration, instead of
// one LDS byte read-modify-write per cell (9604 cells x 17 iterations per 4096 outputs).
//   phase A  waves build the bitmaps row by row with coalesced byte loads + wave ballots
//   phase B  17 synchronous iterations (10 on snow within `area`, 7 on clear within area & water)
//   phase C  waves walk the output rows again (coalesced), finish A11-A15 per pixel
// Window edges are wrong by one more row / column per iteration; after 17 iterations
// exactly the halo is contaminated, so the output region is exact.
// ------------------------------------------------------------------------------
constexpr int CB_W = 128, CB_H = 256, CB_HALO = 17, CB_OUT_W = CB_W - 2 * CB_HALO, CB_OUT_H = CB_H - 2 * CB_HALO;

__global__ __launch_bounds__(256) void dswx_cover_stage2_bits(const KArgs a) {
    typedef unsigned long long u64;
    __shared__ u64 s_init[CB_H][8];          // per row: snow, area, area & water, clear0  (lo, hi each)
    __shared__ u64 s_x[2][CB_H + 2][2];      // row exchange, double-buffered, zero guard rows
    const int H = a.height, W = a.width;
    const long long tile_base = (long long)blockIdx.z * a.tile_stride;
    const int y0 = blockIdx.y * CB_OUT_H - CB_HALO, x0 = blockIdx.x * CB_OUT_W - CB_HALO;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = threadIdx.x;
    if (t < 4) { s_x[t >> 1][(t & 1) ? CB_H + 1 : 0][0] = 0; s_x[t >> 1][(t & 1) ? CB_H + 1 : 0][1] = 0; }
    // A11-A15 as a table over (WTR-2 code, preliminary CLOUD bits, snow): filled once per block by
    // finish_px itself, looked up per pixel in phase C
    __shared__ uint32_t s_fin[128];          // WTR | BWTR << 8 | CONF << 16 | CLOUD << 24
    __shared__ uint8_t s_fbr[128];           // browse
    if (t < 128) {
        const uint32_t c = t & 7u, b = (t >> 3) & 7u;
        PxOut o;
        finish_px(a.P, c < 5u ? c : (c == 5u ? 254u : 255u), (b & 1u) | ((b & 6u) << 1), (t >> 6) != 0, o);
        s_fin[t] = o.wtr | o.bwtr << 8 | o.conf << 16 | o.cloud << 24;
        s_fbr[t] = (uint8_t)o.browse;
    }
    // ---- phase A: four rows per wave and iteration, all 24 byte loads issued before the first
    // ballot consumes one (the loop is latency-bound, not bandwidth-bound)
    const uint8_t* __restrict__ g_fm = a.in.fmask + tile_base;
    const uint8_t* __restrict__ g_pc = a.cover_pc + tile_base;
    const uint8_t* __restrict__ g_w2 = a.cover_w2 + tile_base;
    for (int r0 = wave; r0 < CB_H; r0 += 16) {
        uint32_t fm[4][2], pc[4][2], w2[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + r0 + 4 * j;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int x = x0 + 64 * h + lane;
                const bool in = (y >= 0) & (y < H) & (x >= 0) & (x < W);
                const long long off = in ? (long long)y * W + x : 0;
                fm[j][h] = in ? g_fm[off] : 0u;          // 0 / 1 / 0: no snow, not clear, no water
                pc[j][h] = in ? g_pc[off] : 1u;
                w2[j][h] = in ? g_w2[off] : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t bits[2];                     // snow | area << 1 | (area & water) << 2 | clear0 << 3
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t clear0 = pc[j][h] == 0u ? 1u : 0u, area = ((fm[j][h] >> 2) & 1u) & clear0;
                const uint32_t water = (w2[j][h] - 1u) <= 3u ? 1u : 0u;
                bits[h] = ((fm[j][h] >> 4) & 1u) | area << 1 | (area & water) << 2 | clear0 << 3;
            }
            u64 m[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[2 * k] = __ballot((bits[0] >> k) & 1u);
                m[2 * k + 1] = __ballot((bits[1] >> k) & 1u);
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s_init[r0 + 4 * j][k] = m[k];
            }
        }
    }
    __syncthreads();
    // ---- phase B: thread t owns window row t
    u64 slo = s_init[t][0], shi = s_init[t][1];
    const u64 alo = s_init[t][2], ahi = s_init[t][3], wlo = s_init[t][4], whi = s_init[t][5];
    const u64 c0lo = s_init[t][6], c0hi = s_init[t][7];
    int buf = 0;
    auto step = [&](u64& lo, u64& hi, u64 mlo, u64 mhi) {
        s_x[buf][t + 1][0] = lo; s_x[buf][t + 1][1] = hi;
        __syncthreads();
        const u64 nlo = lo | s_x[buf][t][0] | s_x[buf][t + 2][0] | (lo << 1) | (lo >> 1) | (hi << 63);
        const u64 nhi = hi | s_x[buf][t][1] | s_x[buf][t + 2][1] | (hi << 1) | (hi >> 1) | (lo >> 63);
        lo |= nlo & mlo; hi |= nhi & mhi;
        buf ^= 1;
    };
    for (int it = 0; it < 10; ++it) step(slo, shi, alo, ahi);
    u64 clo = ~slo & c0lo, chi = ~shi & c0hi;
    for (int it = 0; it < 7; ++it) step(clo, chi, wlo, whi);
    // final snow of the row -> LDS (buffer `buf` was last written two steps ago: free)
    s_x[buf][t + 1][0] = slo & ~clo; s_x[buf][t + 1][1] = shi & ~chi;
    __syncthreads();
    // ---- phase C: again four rows per wave and iteration with the loads hoisted
    for (int r0 = CB_HALO + wave; r0 < CB_H - CB_HALO; r0 += 16) {
        uint32_t w2[4][2], pc[4][2];
        bool on[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + 4 * j, y = y0 + r;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = 64 * h + lane, x = x0 + c;
                on[j][h] = (r < CB_H - CB_HALO) & (y < H) & (c >= CB_HALO) & (c < CB_W - CB_HALO) & (x < W);
                const long long off = on[j][h] ? (long long)y * W + x : 0;
                w2[j][h] = on[j][h] ? g_w2[off] : 0u;
                pc[j][h] = on[j][h] ? g_pc[off] : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + 4 * j, y = y0 + r;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!on[j][h]) continue;
                const uint32_t snow = (uint32_t)(s_x[buf][r + 1][h] >> lane) & 1u;
                const long long off = tile_base + (long long)y * W + x0 + 64 * h + lane;
                const uint32_t idx = (w2[j][h] < 5u ? w2[j][h] : w2[j][h] - 249u) |
                                     ((pc[j][h] & 1u) | ((pc[j][h] >> 1) & 6u)) << 3 | snow << 6;
                const uint32_t e = s_fin[idx];
                if (a.out.wtr) a.out.wtr[off] = (uint8_t)e;
                if (a.out.bwtr) a.out.bwtr[off] = (uint8_t)(e >> 8);
                if (a.out.conf) a.out.conf[off] = (uint8_t)(e >> 16);
                if (a.out.cloud) a.out.cloud[off] = (uint8_t)(e >> 24);
                if (a.out.browse) a.out.browse[off] = s_fbr[idx];
            }
        }
    }
}

// ------------------------------------------------------------------------------
// 'cover' mode, stage 2, bit-packed, FOUR pixels per lane (default when rows keep 4-byte
// alignment: width % 4 == 0, tile stride % 4 == 0, 4-byte aligned planes).  Same window scheme as
// dswx_cover_stage2_bits (128 x 256, one thread = one window row in phase B) with a column halo
// of 20 so that every lane's quad is dword-aligned (88 x 222 outputs per block):
//   phase A  a wave takes TWO rows per step (half-wave each), one dword load per plane and lane,
//            byte-parallel predicates, ballots -> the row bitmap as 4 x u32 in pixel-interleaved
//            order: word k, bit l <-> window column 4 l + k
//   phase B  in that order the horizontal neighbours are plain word moves:
//            left(k) = word k-1 (k > 0), word 3 << 1 (k = 0);  right(k) = word k+1, word 0 >> 1
//   phase C  dword loads, four table lookups, byte transpose, dword stores
// A quarter of the memory instructions and ballots of the byte-lane version per pixel.
// ------------------------------------------------------------------------------
constexpr int CQ_HALO_X = 20, CQ_OUT_W = CB_W - 2 * CQ_HALO_X;

__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {      // 0x01 in every byte of v that is 0
    return (~(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v | 0x7f7f7f7fu)) >> 7;
}

__global__ __launch_bounds__(256) void dswx_cover_stage2_quads(const KArgs a) {
    __shared__ uint32_t s_init[CB_H][16];     // per row: snow[4], area[4], area & water[4], clear0[4]
    __shared__ uint32_t s_x[2][CB_H + 2][4];  // row exchange, double-buffered, zero guard rows
    __shared__ uint32_t s_fin[128];           // WTR | BWTR << 8 | CONF << 16 | CLOUD << 24
    __shared__ uint8_t s_fbr[128];            // browse
    const int H = a.height, W = a.width;
    const long long tile_base = (long long)blockIdx.z * a.tile_stride;
    const int y0 = blockIdx.y * CB_OUT_H - CB_HALO, x0 = blockIdx.x * CQ_OUT_W - CQ_HALO_X;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = threadIdx.x;
    const int l32 = lane & 31, rsel = lane >> 5;
    if (t < 16) s_x[t >> 3][((t >> 2) & 1) ? CB_H + 1 : 0][t & 3] = 0;
    if (t < 128) {
        const uint32_t c = t & 7u, b = (t >> 3) & 7u;
        PxOut o;
        finish_px(a.P, c < 5u ? c : (c == 5u ? 254u : 255u), (b & 1u) | ((b & 6u) << 1), (t >> 6) != 0, o);
        s_fin[t] = o.wtr | o.bwtr << 8 | o.conf << 16 | o.cloud << 24;
        s_fbr[t] = (uint8_t)o.browse;
    }
    const uint8_t* __restrict__ g_fm = a.in.fmask + tile_base;
    const uint8_t* __restrict__ g_pc = a.cover_pc + tile_base;
    const uint8_t* __restrict__ g_w2 = a.cover_w2 + tile_base;
    const int x = x0 + 4 * l32;
    const bool x_in = (x >= 0) & (x < W);                 // W % 4 == 0: a quad is inside or outside as a whole
    // ---- phase A: row pairs p = wave + 4 i (rows 2p, 2p + 1), four pairs per iteration
    for (int p0 = wave; p0 < CB_H / 2; p0 += 16) {
        uint32_t fm[4], pc[4], w2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + 2 * (p0 + 4 * j) + rsel;
            const bool in = x_in & (y >= 0) & (y < H);
            const long long off = in ? (long long)y * W + x : 0;
            fm[j] = in ? *reinterpret_cast<const uint32_t*>(g_fm + off) : 0u;
            pc[j] = in ? *reinterpret_cast<const uint32_t*>(g_pc + off) : 0x01010101u;   // not clear
            w2[j] = in ? *reinterpret_cast<const uint32_t*>(g_w2 + off) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t snow4 = (fm[j] >> 4) & 0x01010101u, clear4 = zero_bytes(pc[j]);
            const uint32_t area4 = (fm[j] >> 2) & clear4;                       // clear4 is 0 / 1 per byte
            // WTR-2 uncollapsed is one of 0..4, 254, 255: water classes are the nonzero bytes below 8
            const uint32_t water4 = zero_bytes(w2[j] & 0xf8f8f8f8u) & ~zero_bytes(w2[j]);
            const uint32_t m4[4] = {snow4, area4, area4 & water4, clear4};
            const int row = 2 * (p0 + 4 * j);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long bal = __ballot((m4[q] >> (8 * k)) & 1u);
                    if (lane == 0) s_init[row][4 * q + k] = (uint32_t)bal;
                    if (lane == 1) s_init[row + 1][4 * q + k] = (uint32_t)(bal >> 32);
                }
            }
        }
    }
    __syncthreads();
    // ---- phase B: thread t owns window row t
    uint32_t S[4], A[4], Wm[4], C0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { S[k] = s_init[t][k]; A[k] = s_init[t][4 + k]; Wm[k] = s_init[t][8 + k]; C0[k] = s_init[t][12 + k]; }
    int buf = 0;
    auto step = [&](uint32_t (&X)[4], const uint32_t (&M)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_x[buf][t + 1][k] = X[k];
        __syncthreads();
        uint32_t n[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) n[k] = X[k] | s_x[buf][t][k] | s_x[buf][t + 2][k];
        n[0] |= (X[3] << 1) | X[1];
        n[1] |= X[0] | X[2];
        n[2] |= X[1] | X[3];
        n[3] |= X[2] | (X[0] >> 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] |= n[k] & M[k];
        buf ^= 1;
    };
    for (int it = 0; it < 10; ++it) step(S, A);
    uint32_t C[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) C[k] = ~S[k] & C0[k];
    for (int it = 0; it < 7; ++it) step(C, Wm);
#pragma unroll
    for (int k = 0; k < 4; ++k) s_x[buf][t + 1][k] = S[k] & ~C[k];      // final snow of the row
    __syncthreads();
    // ---- phase C: row pairs again; output columns 20..107 = quads 5..26
    const bool x_out = x_in & (l32 >= CQ_HALO_X / 4) & (l32 < (CB_W - CQ_HALO_X) / 4);
    for (int p0 = wave; p0 < CB_H / 2; p0 += 16) {
        uint32_t w2[4], pc[4];
        bool on[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 2 * (p0 + 4 * j) + rsel, y = y0 + r;
            on[j] = x_out & (r >= CB_HALO) & (r < CB_H - CB_HALO) & (y < H);
            const long long off = on[j] ? (long long)y * W + x : 0;
            w2[j] = on[j] ? *reinterpret_cast<const uint32_t*>(g_w2 + off) : 0u;
            pc[j] = on[j] ? *reinterpret_cast<const uint32_t*>(g_pc + off) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!on[j]) continue;
            const int r = 2 * (p0 + 4 * j) + rsel;
            uint32_t e[4], br = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t snow = (s_x[buf][r + 1][k] >> l32) & 1u;
                const uint32_t w = (w2[j] >> (8 * k)) & 0xffu, c = (pc[j] >> (8 * k)) & 0xffu;
                const uint32_t idx = (w < 5u ? w : w - 249u) | ((c & 1u) | ((c >> 1) & 6u)) << 3 | snow << 6;
                e[k] = s_fin[idx];
                br |= (uint32_t)s_fbr[idx] << (8 * k);
            }
            uint32_t planes[4];            // byte k of every e -> plane k, pixel order
            transpose4(e, planes);
            const long long off = tile_base + (long long)(y0 + r) * W + x;
            if (a.out.wtr) *reinterpret_cast<uint32_t*>(a.out.wtr + off) = planes[0];
            if (a.out.bwtr) *reinterpret_cast<uint32_t*>(a.out.bwtr + off) = planes[1];
            if (a.out.conf) *reinterpret_cast<uint32_t*>(a.out.conf + off) = planes[2];
            if (a.out.cloud) *reinterpret_cast<uint32_t*>(a.out.cloud + off) = planes[3];
            if (a.out.browse) *reinterpret_cast<uint32_t*>(a.out.browse + off) = br;
        }
    }
}

// Sums the fused kernel's per-wave partial counts of one tile (block = tile) and
// WRITES counters[tile]; the ragged-remainder kernel adds to them afterwards.
__global__ __launch_bounds__(1024) void dswx_counters_finish(const uint2* __restrict__ partials,
                                                             unsigned long long* __restrict__ counters,
                                                             long long per_tile, int has_ocean,
                                                             long long vec_pixels) {
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
        if (threadIdx.x == 2 && !has_ocean) sum = (unsigned long long)vec_pixels;
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
    uint32_t c0 = 0, c1 = 0, c2 = 0;
    const long long px = a.px_begin + (long long)blockIdx.x * 256 + threadIdx.x;
    if (px < a.n_pixels) {
        const long long off = (long long)blockIdx.y * a.tile_stride + px;
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
    const int cls_of_row[4] = {0, 2, 3, 4};
    for (int v = 0; v < 256 && p->apply_aerosol_class_remapping; ++v) {
        uint32_t bits = 0;
        for (int k = 0; k < 4; ++k)
            if (p->aerosol_fmask_lut[k][v]) bits |= 1u << cls_of_row[k];
        d->aer_lut[v >> 2] |= bits << (8 * (v & 3));
    }
    return DSWX_OK;
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
    if (!p) return dswx_fail(DSWX_ERR_ARG, "params is NULL");
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
    if (!out) return dswx_fail(DSWX_ERR_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return dswx_fail(DSWX_ERR_NO_DEVICE, "no HIP device visible: the DSWx HIP path has no CPU fallback");
    if (device < 0 || device >= n) return dswx_fail(DSWX_ERR_ARG, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    dswx_ctx* c = new dswx_ctx();
    c->device = device;
    if (const char* e = std::getenv("DSWX_FUSED_VARIANT")) {
        const int v = std::atoi(e);
        c->fused_variant = (v >= 0 && v <= 5) ? v : -1;
    }
    if (const char* e = std::getenv("DSWX_TUNE_WPS")) c->tune_wps = std::atoi(e);
    if (const char* e = std::getenv("DSWX_TUNE_ABLATE")) c->tune_ablate = std::atoi(e);
    if (const char* e = std::getenv("DSWX_TUNE_PIPE_BLOCKS")) c->tune_pipe_blocks = std::atoi(e);
    if (const char* e = std::getenv("DSWX_TUNE_LUT_WPS")) c->tune_lut_wps = std::atoi(e);
    if (const char* e = std::getenv("DSWX_COVER_KERNEL")) c->cover_kernel = std::atoi(e);
    if (const char* e = std::getenv("DSWX_HOST_PIPELINE")) c->host_pipeline = std::atoi(e);
    if (const char* e = std::getenv("DSWX_HOST_CHUNKS")) { const int v = std::atoi(e); if (v >= 1 && v <= 256) c->host_chunks = v; }
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
    if (ctx->cover) (void)hipFree(ctx->cover);
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
    delete ctx;
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
    KArgs a;
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
    a.in = *in;
    a.out = *out;
    a.counters = reinterpret_cast<unsigned long long*>(counters);
    a.n_pixels = n_pixels;
    a.tile_stride = tile_stride;
    a.cover_w2 = a.cover_pc = nullptr;
    a.height = (int)height; a.width = (int)width;
    dswx_planes_out_t final_out = *out;     // what stage 2 of 'cover' writes
    if (cover) {
        const size_t need = 2 * (size_t)n_tiles * (size_t)tile_stride;
        if (need > ctx->cover_bytes) {
            HIP_TRY(hipStreamSynchronize(s));
            if (ctx->cover) HIP_TRY(hipFree(ctx->cover));
            ctx->cover = nullptr; ctx->cover_bytes = 0;
            HIP_TRY(hipMalloc(&ctx->cover, need));
            ctx->cover_bytes = need;
        }
        a.cover_w2 = static_cast<uint8_t*>(ctx->cover);
        a.cover_pc = a.cover_w2 + (size_t)n_tiles * (size_t)tile_stride;
        // stage 1 stops before the snow step: these four layers come from stage 2
        a.out.wtr = a.out.bwtr = a.out.conf = a.out.cloud = a.out.browse = nullptr;
    }

    const bool any_index = out->mndwi || out->ndvi || out->awesh;
    const bool masks = in->land || in->shad || in->ocean;
    // the fused kernel needs every plane 16-byte aligned at every tile start
    bool vec_ok = (tile_stride % 16 == 0) || n_tiles == 1;
    for (int k = 0; k < 6 && vec_ok; ++k) vec_ok = aligned_to(in->band[k], 16);
    vec_ok = vec_ok && aligned_to(in->fmask, 16) && (!in->land || aligned_to(in->land, 16)) &&
             (!in->shad || aligned_to(in->shad, 16)) && (!in->ocean || aligned_to(in->ocean, 16)) &&
             (!out->diag || aligned_to(out->diag, 16));
    uint8_t* const u8outs[] = {out->wtr1, out->wtr1_aerosol, out->wtr2, out->wtr, out->bwtr, out->conf, out->cloud,
                               out->browse};
    for (uint8_t* p : u8outs) vec_ok = vec_ok && (!p || aligned_to(p, 16));
    // are all tile starts of all planes on 256-byte boundaries?
    bool aligned256 = (tile_stride % 256 == 0) || n_tiles == 1;
    for (int k = 0; k < 6 && aligned256; ++k) aligned256 = aligned_to(in->band[k], 256);
    aligned256 = aligned256 && aligned_to(in->fmask, 256) && (!in->land || aligned_to(in->land, 256)) &&
                 (!in->shad || aligned_to(in->shad, 256)) && (!in->ocean || aligned_to(in->ocean, 256)) &&
                 (!out->diag || aligned_to(out->diag, 256));
    for (uint8_t* p : u8outs) aligned256 = aligned256 && (!p || aligned_to(p, 256));

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
        if (b.cover_w2) { b.cover_w2 += shift; b.cover_pc += shift; }
        b.px_begin = 0;
        b.partials = nullptr;
        const int64_t groups = vec_ok ? (n_pixels >> 3) : 0;
        // the finishing kernel of the vector path WRITES the counters; only the generic kernel
        // alone (atomic adds) needs them zeroed first
        if (groups == 0 && b.counters)
            HIP_TRY(hipMemsetAsync(b.counters, 0, (size_t)nt * 3 * sizeof(int64_t), s));
        if (groups > 0) {
            // 'cover' stage 1 and the browse plane: the direct kernel or the table-driven one (3), not
            // the experimental structures
            const bool plain_outputs = !cover && !b.out.browse;
            // automatic choice: the table-driven kernel when every tile of every plane starts on
            // a 256-byte boundary (6.1 vs 5.5 TB/s there), the direct kernel otherwise (5.4 vs 5.1)
            int vsel = ctx->fused_variant;
            if (vsel < 0) vsel = aligned256 ? 3 : 0;
            // the LDS-DMA variants (2, 4, 5) move 16 pixels per lane of the u8 planes
            const bool dma16_ok = (n_pixels & 15) == 0 || vsel == 1 || vsel == 3;
            const bool variant = vsel != 0 && (plain_outputs || vsel == 3) && dma16_ok;
            int threads = 256;
            long long gx_ll = (groups + 255) / 256;
            if (variant && vsel == 3) dswx_lut_geometry(ctx, groups, &threads, &gx_ll);
            else if (variant) dswx_variant_geometry(ctx, vsel, groups, nt, &threads, &gx_ll);
            const int64_t gx = gx_ll;
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
            if (variant) {
                const int vrc = vsel == 3 ? dswx_lut_launch(ctx, b, masks, grid, block, s, info, sizeof info)
                                          : dswx_variant_launch(ctx, vsel, b, masks, grid, block, s, info, sizeof info);
                if (vrc) return vrc;
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
                hipLaunchKernelGGL(dswx_counters_finish, dim3((unsigned)nt), dim3(1024), 0, s, b.partials,
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
            // stage-2 kernel: 2 = bitmaps with four pixels per lane (needs dword-aligned rows),
            // 1 = bitmaps with one pixel per lane, 0 = byte cells; env DSWX_COVER_KERNEL caps the choice
            int ck = ctx->cover_kernel;
            if (ck >= 2) {
                bool quad_ok = width % 4 == 0 && (tile_stride % 4 == 0 || n_tiles == 1) && aligned_to(c2.in.fmask, 4) &&
                               aligned_to(c2.cover_w2, 4) && aligned_to(c2.cover_pc, 4);
                uint8_t* const outs[5] = {c2.out.wtr, c2.out.bwtr, c2.out.conf, c2.out.cloud, c2.out.browse};
                for (uint8_t* o : outs) quad_ok = quad_ok && (!o || aligned_to(o, 4));
                ck = quad_ok ? 2 : 1;
            }
            const int tw = ck == 2 ? CQ_OUT_W : (ck == 1 ? CB_OUT_W : CV_TILE), th = ck ? CB_OUT_H : CV_TILE;
            dim3 grid((unsigned)((width + tw - 1) / tw), (unsigned)((height + th - 1) / th), (unsigned)nt);
            if (grid.y > 65535) return dswx_fail(DSWX_ERR_ARG, "raster too tall for one launch");
            if (c2.out.wtr || c2.out.bwtr || c2.out.conf || c2.out.cloud || c2.out.browse) {
                if (ck == 2) hipLaunchKernelGGL(dswx_cover_stage2_quads, grid, dim3(256), 0, s, c2);
                else if (ck == 1) hipLaunchKernelGGL(dswx_cover_stage2_bits, grid, dim3(256), 0, s, c2);
                else hipLaunchKernelGGL(dswx_cover_stage2, grid, dim3(256), 0, s, c2);
                HIP_TRY(hipGetLastError());
            }
            const size_t len = strlen(info);
            snprintf(info + len, sizeof info - len, " + %s grid=(%u,%u,%u)",
                     ck == 2 ? "dswx_cover_stage2_quads" : (ck == 1 ? "dswx_cover_stage2_bits" : "dswx_cover_stage2"),
                     grid.x, grid.y, grid.z);
        }
        if (any_index) {
            dim3 grid((unsigned)((n_pixels + 255) / 256), (unsigned)nt), block(256);
            hipLaunchKernelGGL(dswx_indices_v1, grid, block, 0, s, b);
            HIP_TRY(hipGetLastError());
        }
    }
    ctx->last_kernel = info;
    return DSWX_OK;
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
    // DSWX_MALLOC_FLAGS (experiments): hipExtMallocWithFlags flags, e.g. 4 = hipDeviceMallocContiguous
    static const int flags = [] { const char* e = std::getenv("DSWX_MALLOC_FLAGS"); return e ? std::atoi(e) : 0; }();
    if (flags) {
        const hipError_t e = hipExtMallocWithFlags(out, bytes ? bytes : 1, (unsigned)flags);
        if (e == hipSuccess) return DSWX_OK;
        (void)hipGetLastError();          // fall back to the default allocator
    }
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
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
