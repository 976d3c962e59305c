// dswx_cover.hip -- mask_adjacent_to_cloud_mode 'cover' (SURVEY.md row f2), stage 2:
// _add_snow_to_cloud_layer :2055-2078, then A11-A15.
//   snow  = dilate^10(Fmask bit 4)            restricted to  area = adjacent & (CLOUD == 0)
//   clear = dilate^7(~snow & (CLOUD == 0))    restricted to  area & (WTR-2 in 1..4)
//   snow &= ~clear
// with scipy.ndimage.binary_dilation semantics: 4-neighbour cross, synchronous iterations, cells
// outside the mask keep their value, outside the raster = False.
//
// Stage 1 is the fused classifier itself (EXTRAS instantiation), which stops before the snow step
// and parks ONE byte per pixel in HBM, the "cover state" (cover_state_of in dswx_device.h):
//     bits 0-2 WTR-2 code | bit 3-5 CLOUD bits 0, 2, 3 before the snow step | bit 6 Fmask snow | bit 7 Fmask adjacent
// -- everything stage 2 needs: the four dilation predicates are bit tests on it, and its low six bits
// plus the DILATED snow bit index the 128-entry table that finish_px (A11-A15) fills.  (Round 1 parked
// WTR-2 and CLOUD as two planes and re-read Fmask: 2 B written + 3 B x halo read per pixel; now 1 + 1.)
//
// Stage 2, bit-packed: the predicates live as BITMAPS.  One block owns a 128-column x 256-row window
// (halo 17 rows; 17 or 20 columns), one thread owns one window row per mask.  A masked 4-neighbour
// dilation step is then
//     x |= (x | left | right | row_above | row_below) & mask
// on 128 bits: ~30 VALU and one 16-byte LDS exchange per thread and iteration.
//   phase A  waves build the bitmaps row by row with coalesced loads + wave ballots
//   phase B  up to 10 + 7 synchronous iterations; a dilation that changed nothing in the whole window has
//            converged (every later iteration is the identity), so the loop stops there -- the flag rides
//            on the barrier the row exchange needs anyway
//   phase C  waves walk the output rows again (coalesced), finish A11-A15 per pixel from the table
// Window edges are wrong by one more row / column per iteration; after 17 iterations exactly the
// halo is contaminated, so the output region is exact.
#include <hip/hip_runtime.h>

#include "dswx_host.h"
#include "dswx_tables.h"     // transpose4 (byte transposes)

constexpr int CB_W = 128, CB_H = 256, CB_HALO = 17, CB_OUT_W = CB_W - 2 * CB_HALO, CB_OUT_H = CB_H - 2 * CB_HALO;
constexpr uint32_t CV_STATE_OUTSIDE = 0x08u;     // outside the raster: no snow, not adjacent, CLOUD != 0, not water

// A11-A15 as a table over (cover state bits 0-5, dilated snow): filled once per block by finish_px itself
__device__ __forceinline__ void cover_fin_table(const DevParams& P, int t, uint32_t* s_fin, uint8_t* s_fbr) {
    if (t < 128) {
        const uint32_t c = t & 7u, b = (t >> 3) & 7u;
        PxOut o;
        finish_px(P, c < 5u ? c : (c == 5u ? 254u : 255u), (b & 1u) | ((b & 6u) << 1), (t >> 6) != 0, o);
        s_fin[t] = o.wtr | o.bwtr << 8 | o.conf << 16 | o.cloud << 24;
        s_fbr[t] = (uint8_t)o.browse;
    }
}

// block-wide "did any thread change a bit" riding on the exchange barrier: every wave posts its vote
// before the barrier, every thread reads the four votes after it
__device__ __forceinline__ void post_vote(uint32_t (*s_vote)[4], int buf, bool changed) {
    const unsigned long long b = __ballot(changed);
    if ((threadIdx.x & 63) == 0) s_vote[buf][threadIdx.x >> 6] = b != 0ull ? 1u : 0u;
}
__device__ __forceinline__ bool any_vote(uint32_t (*s_vote)[4], int buf) {
    return (s_vote[buf][0] | s_vote[buf][1] | s_vote[buf][2] | s_vote[buf][3]) != 0u;
}

// ------------------------------------------------------------------------------
// One pixel per lane (any width / alignment): 94 x 222 outputs per block.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dswx_cover_stage2_bits(const KArgs a) {
    typedef unsigned long long u64;
    __shared__ u64 s_init[CB_H][8];          // per row: snow, area, area & water, clear0  (lo, hi each)
    __shared__ u64 s_x[2][CB_H + 2][2];      // row exchange, double-buffered, zero guard rows
    __shared__ uint32_t s_vote[2][4];
    __shared__ uint32_t s_fin[128];          // WTR | BWTR << 8 | CONF << 16 | CLOUD << 24
    __shared__ uint8_t s_fbr[128];           // browse
    const int H = a.height, W = a.width;
    const long long tile_base = (long long)blockIdx.z * a.tile_stride;
    const int y0 = blockIdx.y * CB_OUT_H - CB_HALO, x0 = blockIdx.x * CB_OUT_W - CB_HALO;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = threadIdx.x;
    if (t < 4) { s_x[t >> 1][(t & 1) ? CB_H + 1 : 0][0] = 0; s_x[t >> 1][(t & 1) ? CB_H + 1 : 0][1] = 0; }
    cover_fin_table(a.P, t, s_fin, s_fbr);
    // ---- phase A: four rows per wave and iteration, all loads issued before the first ballot
    const uint8_t* __restrict__ g_st = a.cover_state + tile_base;
    for (int r0 = wave; r0 < CB_H; r0 += 16) {
        uint32_t st[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + r0 + 4 * j;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int x = x0 + 64 * h + lane;
                const bool in = (y >= 0) & (y < H) & (x >= 0) & (x < W);
                st[j][h] = in ? g_st[(long long)y * W + x] : CV_STATE_OUTSIDE;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u64 m[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t s = st[j][h];
                const bool clear0 = (s & 0x38u) == 0u, area = clear0 & ((s & 0x80u) != 0u);
                const bool water = ((s & 7u) - 1u) <= 3u;
                m[0 + h] = __ballot((s & 0x40u) != 0u);
                m[2 + h] = __ballot(area);
                m[4 + h] = __ballot(area & water);
                m[6 + h] = __ballot(clear0);
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
    auto dilate = [&](u64& lo, u64& hi, u64 mlo, u64 mhi, int iterations) {
        bool changed = true;
        for (int it = 0; it < iterations; ++it) {
            s_x[buf][t + 1][0] = lo; s_x[buf][t + 1][1] = hi;
            post_vote(s_vote, buf, changed);
            __syncthreads();
            if (!any_vote(s_vote, buf)) { buf ^= 1; break; }        // the previous step was the identity: converged
            const u64 nlo = lo | s_x[buf][t][0] | s_x[buf][t + 2][0] | (lo << 1) | (lo >> 1) | (hi << 63);
            const u64 nhi = hi | s_x[buf][t][1] | s_x[buf][t + 2][1] | (hi << 1) | (hi >> 1) | (lo >> 63);
            const u64 add_lo = nlo & mlo & ~lo, add_hi = nhi & mhi & ~hi;
            changed = (add_lo | add_hi) != 0ull;
            lo |= add_lo; hi |= add_hi;
            buf ^= 1;
        }
    };
    dilate(slo, shi, alo, ahi, 10);
    u64 clo = ~slo & c0lo, chi = ~shi & c0hi;
    dilate(clo, chi, wlo, whi, 7);
    // final snow of the row -> LDS (buffer `buf` was last read two barriers ago: free)
    s_x[buf][t + 1][0] = slo & ~clo; s_x[buf][t + 1][1] = shi & ~chi;
    __syncthreads();
    // ---- phase C: again four rows per wave and iteration with the loads hoisted
    for (int r0 = CB_HALO + wave; r0 < CB_H - CB_HALO; r0 += 16) {
        uint32_t st[4][2];
        bool on[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + 4 * j, y = y0 + r;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = 64 * h + lane, x = x0 + c;
                on[j][h] = (r < CB_H - CB_HALO) & (y < H) & (c >= CB_HALO) & (c < CB_W - CB_HALO) & (x < W);
                st[j][h] = on[j][h] ? g_st[(long long)y * W + x] : 0u;
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
                const uint32_t idx = (st[j][h] & 0x3fu) | snow << 6;
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
// FOUR pixels per lane (default when rows keep 4-byte alignment: width % 4 == 0, tile stride % 4 == 0,
// 4-byte aligned planes).  Same window scheme with a column halo of 20 so that every lane's quad is
// dword-aligned (88 x 222 outputs per block):
//   phase A  a wave takes TWO rows per step (half-wave each), one dword of cover state per lane,
//            byte-parallel predicates, ballots -> the row bitmap as 4 x u32 in pixel-interleaved
//            order: word k, bit l <-> window column 4 l + k
//   phase B  in that order the horizontal neighbours are plain word moves:
//            left(k) = word k-1 (k > 0), word 3 << 1 (k = 0);  right(k) = word k+1, word 0 >> 1
//   phase C  dword loads, four table lookups, byte transpose, dword stores
// ------------------------------------------------------------------------------
constexpr int CQ_HALO_X = 20, CQ_OUT_W = CB_W - 2 * CQ_HALO_X;

__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {      // 0x01 in every byte of v that is 0
    return (~(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v | 0x7f7f7f7fu)) >> 7;
}

__global__ __launch_bounds__(256) void dswx_cover_stage2_quads(const KArgs a) {
    __shared__ uint32_t s_init[CB_H][16];     // per row: snow[4], area[4], area & water[4], clear0[4]
    __shared__ uint32_t s_x[2][CB_H + 2][4];  // row exchange, double-buffered, zero guard rows
    __shared__ uint32_t s_vote[2][4];
    __shared__ uint32_t s_fin[128];           // WTR | BWTR << 8 | CONF << 16 | CLOUD << 24
    __shared__ uint8_t s_fbr[128];            // browse
    const int H = a.height, W = a.width;
    const long long tile_base = (long long)blockIdx.z * a.tile_stride;
    const int y0 = blockIdx.y * CB_OUT_H - CB_HALO, x0 = blockIdx.x * CQ_OUT_W - CQ_HALO_X;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = threadIdx.x;
    const int l32 = lane & 31, rsel = lane >> 5;
    if (t < 16) s_x[t >> 3][((t >> 2) & 1) ? CB_H + 1 : 0][t & 3] = 0;
    cover_fin_table(a.P, t, s_fin, s_fbr);
    const uint8_t* __restrict__ g_st = a.cover_state + tile_base;
    const int x = x0 + 4 * l32;
    const bool x_in = (x >= 0) & (x < W);                 // W % 4 == 0: a quad is inside or outside as a whole
    // ---- phase A: row pairs p = wave + 4 i (rows 2p, 2p + 1), eight pairs per iteration
    for (int p0 = wave; p0 < CB_H / 2; p0 += 32) {
        uint32_t st[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int y = y0 + 2 * (p0 + 4 * j) + rsel;
            const bool in = x_in & (y >= 0) & (y < H);
            st[j] = in ? *reinterpret_cast<const uint32_t*>(g_st + (long long)y * W + x) : CV_STATE_OUTSIDE * 0x01010101u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t s = st[j];
            const uint32_t snow4 = (s >> 6) & 0x01010101u, clear4 = zero_bytes(s & 0x38383838u);
            const uint32_t area4 = (s >> 7) & clear4;                                  // clear4 is 0 / 1 per byte
            const uint32_t code4 = s & 0x07070707u;
            // water classes: code 1..4  <=>  nonzero and (code + 3) has bit 3 clear
            const uint32_t water4 = ~zero_bytes(code4) & ~((code4 + 0x03030303u) >> 3) & 0x01010101u;
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
    auto dilate = [&](uint32_t (&X)[4], const uint32_t (&M)[4], int iterations) {
        bool changed = true;
        for (int it = 0; it < iterations; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_x[buf][t + 1][k] = X[k];
            post_vote(s_vote, buf, changed);
            __syncthreads();
            if (!any_vote(s_vote, buf)) { buf ^= 1; break; }        // the previous step was the identity: converged
            uint32_t n[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) n[k] = X[k] | s_x[buf][t][k] | s_x[buf][t + 2][k];
            n[0] |= (X[3] << 1) | X[1];
            n[1] |= X[0] | X[2];
            n[2] |= X[1] | X[3];
            n[3] |= X[2] | (X[0] >> 1);
            uint32_t grew = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const uint32_t add = n[k] & M[k] & ~X[k]; grew |= add; X[k] |= add; }
            changed = grew != 0u;
            buf ^= 1;
        }
    };
    dilate(S, A, 10);
    uint32_t C[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) C[k] = ~S[k] & C0[k];
    dilate(C, Wm, 7);
#pragma unroll
    for (int k = 0; k < 4; ++k) s_x[buf][t + 1][k] = S[k] & ~C[k];      // final snow of the row
    __syncthreads();
    // ---- phase C: row pairs again; output columns 20..107 = quads 5..26
    const bool x_out = x_in & (l32 >= CQ_HALO_X / 4) & (l32 < (CB_W - CQ_HALO_X) / 4);
    for (int p0 = wave; p0 < CB_H / 2; p0 += 32) {
        uint32_t st[8];
        bool on[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 2 * (p0 + 4 * j) + rsel, y = y0 + r;
            on[j] = x_out & (r >= CB_HALO) & (r < CB_H - CB_HALO) & (y < H);
            st[j] = on[j] ? *reinterpret_cast<const uint32_t*>(g_st + (long long)y * W + x) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (!on[j]) continue;
            const int r = 2 * (p0 + 4 * j) + rsel;
            uint32_t e[4], br = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t snow = (s_x[buf][r + 1][k] >> l32) & 1u;
                const uint32_t idx = ((st[j] >> (8 * k)) & 0x3fu) | snow << 6;
                e[k] = s_fin[idx];
                br |= (uint32_t)s_fbr[idx] << (8 * k);
            }
            uint32_t planes[4];            // byte k of every e -> plane k, pixel order
            transpose4(e, planes);
            const long long off = tile_base + (long long)(y0 + r) * W + x;
            if (a.out.wtr) __builtin_nontemporal_store(planes[0], reinterpret_cast<uint32_t*>(a.out.wtr + off));
            if (a.out.bwtr) __builtin_nontemporal_store(planes[1], reinterpret_cast<uint32_t*>(a.out.bwtr + off));
            if (a.out.conf) __builtin_nontemporal_store(planes[2], reinterpret_cast<uint32_t*>(a.out.conf + off));
            if (a.out.cloud) __builtin_nontemporal_store(planes[3], reinterpret_cast<uint32_t*>(a.out.cloud + off));
            if (a.out.browse) __builtin_nontemporal_store(br, reinterpret_cast<uint32_t*>(a.out.browse + off));
        }
    }
}

// Launches stage 2 for the tiles of `c2` (c2.out = the four / five layers stage 2 produces).
int dswx_cover_stage2_launch(dswx_ctx* ctx, const KArgs& c2, long long n_tiles, long long tile_stride,
                             hipStream_t s, char* info, size_t info_len) {
    const int width = c2.width, height = c2.height;
    // 2 = bitmaps with four pixels per lane (needs dword-aligned rows), 1 = one pixel per lane
    int ck = ctx->cover_kernel;
    if (ck >= 2) {
        bool quad_ok = width % 4 == 0 && (tile_stride % 4 == 0 || n_tiles == 1) && aligned_to(c2.cover_state, 4);
        uint8_t* const outs[5] = {c2.out.wtr, c2.out.bwtr, c2.out.conf, c2.out.cloud, c2.out.browse};
        for (uint8_t* o : outs) quad_ok = quad_ok && (!o || aligned_to(o, 4));
        ck = quad_ok ? 2 : 1;
    } else ck = 1;
    const int tw = ck == 2 ? CQ_OUT_W : CB_OUT_W, th = CB_OUT_H;
    dim3 grid((unsigned)((width + tw - 1) / tw), (unsigned)((height + th - 1) / th), (unsigned)n_tiles);
    if (grid.y > 65535) return dswx_fail(DSWX_ERR_ARG, "raster too tall for one launch");
    if (c2.out.wtr || c2.out.bwtr || c2.out.conf || c2.out.cloud || c2.out.browse) {
        if (ck == 2) hipLaunchKernelGGL(dswx_cover_stage2_quads, grid, dim3(256), 0, s, c2);
        else hipLaunchKernelGGL(dswx_cover_stage2_bits, grid, dim3(256), 0, s, c2);
        HIP_TRY(hipGetLastError());
    }
    snprintf(info, info_len, " + %s grid=(%u,%u,%u)", ck == 2 ? "dswx_cover_stage2_quads" : "dswx_cover_stage2_bits",
             grid.x, grid.y, grid.z);
    return DSWX_OK;
}
