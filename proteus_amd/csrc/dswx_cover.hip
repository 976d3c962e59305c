// dswx_cover.hip -- mask_adjacent_to_cloud_mode 'cover' (SURVEY.md row f2), stages 2 and 3:
// _add_snow_to_cloud_layer :2055-2078, then A11-A15.
//   snow  = dilate^10(Fmask bit 4)            restricted to  area = adjacent & (CLOUD == 0)
//   clear = dilate^7(~snow & (CLOUD == 0))    restricted to  area & (WTR-2 in 1..4)
//   snow &= ~clear
// with scipy.ndimage.binary_dilation semantics: 4-neighbour cross, synchronous iterations, cells
// outside the mask keep their value, outside the raster = False.
//
// Three kernels.  Stage 1 is the fused classifier itself (EXTRAS instantiation): it stops before the snow
// step -- DIAG, WTR-1, WTR-2 are final, the other layers are not written -- and parks in HBM
//     cover state byte (cover_state_of, dswx_device.h): bits 0-2 WTR-2 code | bits 3-5 CLOUD bits 0, 2, 3
//         before the snow step | bit 6 Fmask snow | bit 7 Fmask adjacent                       (1 B / px)
//     bitmap dword of every 8-pixel group (cover_bits_of): [snow8, area8, area-and-water8, clear8]  (0.5 B / px)
// Stage 2 (dswx_cover_dilate) works on the bitmaps only and leaves the FINAL snow decision as one bit per
// pixel (0.125 B / px).  One block owns a 256-row x (32 NW)-column window (halo 17 rows, 20 / 17 columns),
// ONE THREAD owns one window row:
//   load     the thread reads its row's 4 NW + 1 bitmap dwords (16-byte loads), transposes the bytes into
//            the four masks and funnel-shifts them to the window's first column -- no ballots: ~100
//            instructions per 256-pixel row (round 1 built the row bitmaps with 16 wave ballots and 32
//            single-lane LDS writes per 256 pixels: 25 k of the 35 k wave instructions of a window)
//   dilate   x |= (x | left | right | row_above | row_below) & mask, up to 10 + 7 synchronous iterations;
//            the row exchange goes through LDS (word-major: conflict-free; one barrier per iteration).  A
//            dilation that changed nothing in the whole window has converged, so the loop stops there -- the
//            vote rides on that barrier
//   store    snow & ~clear of the output region, OR-ed into the flat bit plane (atomics: the rows of
//            neighbouring windows share dwords)
// Stage 3 (dswx_cover_finish) is a flat streaming kernel like stage 1: state byte + final snow bit ->
// 128-entry table filled by finish_px itself (A11-A15) -> WTR, BWTR, CONF, CLOUD (and browse), 8 pixels
// per thread.  Measured on MI355X this round: rewriting only the pixels whose snow decision changed
// ("patching" layers stage 1 had already written) is NOT cheaper -- 1.7 % of the synthetic tile's pixels
// change, but they touch 88 % of the 128-byte lines of the four planes, i.e. the patch re-reads and
// re-writes the planes at line granularity (0.021 ms per tile whatever the kernel structure).
// Window edges are wrong by one more row / column per iteration; after 17 iterations exactly the halo is
// contaminated, so the output region is exact.
#include <hip/hip_runtime.h>

#include "dswx_host.h"
#include "dswx_tables.h"     // transpose4 (byte transposes)

constexpr int CP_H = 256, CP_HALO = 17, CP_OUT_H = CP_H - 2 * CP_HALO;

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// mask of the bits [lo, hi) of a 32-bit word (either bound may lie outside 0..32)
__device__ __forceinline__ uint32_t bit_range(int lo, int hi) {
    uint32_t m = (hi <= 0 || lo >= 32) ? 0u : 0xffffffffu;
    if (lo > 0 && lo < 32) m &= 0xffffffffu << lo;
    if (hi > 0 && hi < 32) m &= 0xffffffffu >> (32 - hi);
    return m;
}

// NW: words per window row, 4 (128 columns) or 8 (256).  STAGED: the block stages the bitmap rows in LDS
// with coalesced loads (each row's piece read by NW + 1 neighbouring lanes) instead of every thread reading
// its own row from HBM.
template <int NW, bool STAGED>
__global__ __launch_bounds__(256) void dswx_cover_dilate(const KArgs a) {
    constexpr int CP_W = 32 * NW, CP_OUT_W = CP_W - 2 * CP_HALO, NQ = NW + 1;     // NQ: 16-byte loads per row
    // one LDS area, two lives: staging of the bitmap rows -> row exchange of the dilations
    constexpr int ROW_DW = 4 * NQ;
    // the bitmap rows are staged in STAGE_PASSES passes of CP_H / STAGE_PASSES rows, so that the staging area is no
    // larger than the row exchange that re-uses it: 16.5 KB per block instead of 36.9 KB (8 words per row), i.e. six
    // resident blocks per CU (the VGPR limit) instead of four
    constexpr int STAGE_PASSES = NW == 8 ? 2 : 1, STAGE_ROWS = CP_H / STAGE_PASSES;
    constexpr int STAGE_DW = STAGED ? STAGE_ROWS * ROW_DW : 0, XCH_DW = 2 * (CP_H + 2) * NW;     // (CP_H (NW + 1) <= XCH_DW)
    constexpr int RAW_DW = STAGE_DW > XCH_DW ? STAGE_DW : XCH_DW;
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[RAW_DW];
    __shared__ uint32_t s_vote[2][4];
    // word-major: the lanes of a wave (consecutive rows) touch consecutive dwords, i.e. distinct LDS banks; a
    // row-major [row][word] exchange puts every 4th (8th) lane on the same bank: 16-way conflicts on every access
    uint32_t (*s_x)[NW][CP_H + 2] = reinterpret_cast<uint32_t (*)[NW][CP_H + 2]>(s_raw);   // [2][NW][CP_H + 2]
    const int H = a.height, W = a.width;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int y0 = blockIdx.y * CP_OUT_H - CP_HALO, x0 = blockIdx.x * CP_OUT_W - CP_HALO;
    const int y = y0 + t;
    // ---- load: the window's bitmap rows.  Window column c of row y is flat pixel y W + x0 + c of the tile,
    // i.e. bit (f0 & 7) onwards of group dword f0 >> 3; a row needs 4 NW + 1 dwords (NQ 16-byte loads).
    // The loads are UNCONDITIONAL (a branch per load serialises them: nine dependent HBM round trips per row
    // instead of one).  Rows outside the raster read a clamped row and are masked below; a piece may start up
    // to 3 dwords before the tile's first group (x0 < 0 in row 0) or run up to ~35 dwords past its last one
    // (last rows of the last window column): the scratch carries 64 dwords of slack behind the last tile and
    // the state plane in front of the first, and whatever is read there lies in masked columns.
    const uint32_t* __restrict__ bits = a.cover_bits + (long long)blockIdx.z * a.cover_bits_stride;
    auto load_quad = [&](int yy, int q) {
        const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
        const long long g = (((long long)yc * W + x0) >> 3) + 4 * q;
        return *reinterpret_cast<const u32x4_a4*>(bits + g);
    };
    uint32_t d[4 * NQ];
    if (STAGED) {
#pragma unroll
        for (int pass = 0; pass < STAGE_PASSES; ++pass) {
            if (pass) __syncthreads();              // the rows of the previous pass have been taken
            for (int id = t; id < STAGE_ROWS * NQ; id += 256) {
                const int r = id / NQ, q = id - r * NQ;
                const u32x4_a4 v = load_quad(y0 + pass * STAGE_ROWS + r, q);
                *reinterpret_cast<uint4*>(&s_raw[r * ROW_DW + 4 * q]) = make_uint4(v.x, v.y, v.z, v.w);
            }
            __syncthreads();
            if (t / STAGE_ROWS == pass) {           // wave-uniform: STAGE_ROWS is a multiple of 64
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const uint4 v = *reinterpret_cast<const uint4*>(&s_raw[(t - pass * STAGE_ROWS) * ROW_DW + 4 * q]);
                    d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
                }
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const u32x4_a4 v = load_quad(y, q);
            d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
        }
    }
    uint32_t S[NW], A[NW], Wm[NW], C0[NW];
    {
        const bool row_in = (y >= 0) & (y < H);
        const int sh = (int)(((long long)(row_in ? y : 0) * W + x0) & 7);      // the flat index may be negative: & is mod
        // byte q of group dword i -> mask q, byte i: four dwords at a time
        uint32_t m[4][NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            uint32_t planes[4];
            transpose4(&d[4 * q], planes);
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k][q] = planes[k];
        }
        // columns of the window that lie inside the raster: [c_lo, c_hi)
        const int c_lo = x0 < 0 ? -x0 : 0, c_hi = (W - x0) < CP_W ? (W - x0) : CP_W;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const uint32_t cm = row_in ? bit_range(c_lo - 32 * k, c_hi - 32 * k) : 0u;
            S[k] = __builtin_amdgcn_alignbit(m[0][k + 1], m[0][k], sh) & cm;
            A[k] = __builtin_amdgcn_alignbit(m[1][k + 1], m[1][k], sh) & cm;
            Wm[k] = __builtin_amdgcn_alignbit(m[2][k + 1], m[2][k], sh) & cm;
            C0[k] = __builtin_amdgcn_alignbit(m[3][k + 1], m[3][k], sh) & cm;
        }
    }
    uint32_t S0[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) S0[k] = S[k];
    if (STAGED) __syncthreads();        // every row has left the staging area: it becomes the row exchange
    // (the guard-row writes below and the first exchange write are ordered by the barrier inside dilate)
    if (t < 2 * NW) { s_x[0][t >> 1][(t & 1) ? CP_H + 1 : 0] = 0; s_x[1][t >> 1][(t & 1) ? CP_H + 1 : 0] = 0; }
    // ---- dilate: thread t owns window row t
    int buf = 0;
    auto dilate = [&](uint32_t (&X)[NW], const uint32_t (&M)[NW], int iterations) {
        bool changed = true;
        for (int it = 0; it < iterations; ++it) {
#pragma unroll
            for (int k = 0; k < NW; ++k) s_x[buf][k][t + 1] = X[k];
            {   // block-wide "did the previous step change a bit": every wave posts its vote before the barrier
                const unsigned long long b = __ballot(changed);
                if (lane == 0) s_vote[buf][wave] = b != 0ull ? 1u : 0u;
            }
            __syncthreads();
            if ((s_vote[buf][0] | s_vote[buf][1] | s_vote[buf][2] | s_vote[buf][3]) == 0u) { buf ^= 1; break; }
            uint32_t grew = 0;
            uint32_t n[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const uint32_t left = __builtin_amdgcn_alignbit(X[k], k > 0 ? X[k - 1] : 0u, 31);          // pixel x - 1
                const uint32_t right = __builtin_amdgcn_alignbit(k + 1 < NW ? X[k + 1] : 0u, X[k], 1);      // pixel x + 1
                n[k] = (s_x[buf][k][t] | s_x[buf][k][t + 2] | left | right) & M[k] & ~X[k];
                grew |= n[k];
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) X[k] |= n[k];
            changed = grew != 0u;
            buf ^= 1;
        }
    };
    dilate(S, A, 10);
    uint32_t C[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) C[k] = ~S[k] & C0[k];
    dilate(C, Wm, 7);
    // ---- store: final snow of the output region into the flat bit plane (bit i of the tile = pixel i).
    // Window word k of row y starts at flat bit f0 = y W + x0 (+ 32 k): shifted by f0 & 31 it lands in the
    // plane dwords (f0 >> 5) + k and + k + 1.  Neighbouring windows share dwords, hence atomic ORs into a
    // zeroed plane.  STAGED: the rows hand their NW + 1 shifted words to LDS and NW + 1 neighbouring lanes
    // issue one row's atomics (a row per lane means 64 cache lines per wave instruction: measured 22 us of
    // the block's 53).
    uint32_t* __restrict__ plane = a.cover_snow + (long long)blockIdx.z * a.cover_snow_stride;
    const bool row_out = (t >= CP_HALO) & (t < CP_H - CP_HALO) & (y < H);
    uint32_t outw[NW + 1];
    {
        const int s32 = (int)(((long long)y * W + x0) & 31);
        uint32_t prev = 0;
#pragma unroll
        for (int k = 0; k <= NW; ++k) {
            // output columns of word k; columns outside the raster are already zero
            const uint32_t cur = (k < NW && row_out) ? (S[k] & ~C[k]) & bit_range(CP_HALO - 32 * k, CP_W - CP_HALO - 32 * k) : 0u;
            // plane dword (f0 >> 5) + k gets cur << s32 | prev >> (32 - s32)
            const uint32_t v = __builtin_amdgcn_alignbit(cur, prev, (32 - s32) & 31);
            outw[k] = s32 ? v : cur;
            prev = cur;
        }
    }
    if (STAGED) {
        constexpr int OW = NW + 1;                  // odd: consecutive rows start on distinct banks
        __syncthreads();                            // the last row exchange has been read: the area is free again
#pragma unroll
        for (int k = 0; k <= NW; ++k) s_raw[t * OW + k] = outw[k];
        __syncthreads();
        for (int id = t; id < CP_H * OW; id += 256) {
            const int r = id / OW, k = id - r * OW;
            const uint32_t v = s_raw[id];
            if (v) {
                const long long w0 = ((long long)(y0 + r) * W + x0) >> 5;        // v != 0 => the row is inside the raster
                __hip_atomic_fetch_or(plane + (w0 + k), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        const long long w0 = ((long long)y * W + x0) >> 5;
#pragma unroll
        for (int k = 0; k <= NW; ++k)
            if (outw[k]) __hip_atomic_fetch_or(plane + (w0 + k), outw[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------
// Stage 3: A11-A15 of every pixel from its state byte and its final snow bit.  Flat, 8 pixels per thread:
// one 8-byte state load + one byte of the snow plane in, four (five) 8-byte stores out -- unaligned global accesses since
// round 6 (any plane address, any tile stride: ragged contiguous batches used to take byte accesses throughout); the
// incomplete last group of a tile goes byte by byte.
// ------------------------------------------------------------------------------
constexpr int FIN_GROUPS = 4;     // 8-pixel groups per thread, 256 groups apart: four loads in flight per thread

__global__ __launch_bounds__(256) void dswx_cover_finish(const KArgs a) {
    __shared__ uint32_t s_fin[128];             // WTR | BWTR << 8 | CONF << 16 | CLOUD << 24
    __shared__ uint32_t s_fbr[128];             // browse
    const int t = threadIdx.x;
    if (t < 128) {   // over (cover state bits 0-5, final snow), by finish_px itself
        const uint32_t c = t & 7u, b = (t >> 3) & 7u;
        PxOut o;
        finish_px(a.P, c < 5u ? c : (c == 5u ? 254u : 255u), (b & 1u) | ((b & 6u) << 1), (t >> 6) != 0, o);
        s_fin[t] = o.wtr | o.bwtr << 8 | o.conf << 16 | o.cloud << 24;
        s_fbr[t] = o.browse;
    }
    __syncthreads();
    const uint8_t* __restrict__ snow_plane = reinterpret_cast<const uint8_t*>(a.cover_snow + (long long)blockIdx.y * a.cover_snow_stride);
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
    // a thread's state bytes carry ~8 bytes per wave lane: with one group per thread the kernel is bound by
    // waves-in-flight x bytes-per-wave / latency (5.4 TB/s measured); all FIN_GROUPS loads go out first
    uint8_t* const planes[5] = {a.out.wtr, a.out.bwtr, a.out.conf, a.out.cloud, a.out.browse};
    auto finish_group = [&](const uint32_t (&st)[2], uint32_t snow8, uint32_t (&lo)[5], uint32_t (&hi)[5]) {
        uint32_t e[8], br[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t idx = ((st[j >> 2] >> (8 * (j & 3))) & 0x3fu) | ((snow8 >> j) & 1u) << 6;
            e[j] = s_fin[idx];
            br[j >> 2] |= s_fbr[idx] << (8 * (j & 3));
        }
        uint32_t l4[4], h4[4];              // byte k of every e -> plane k, pixel order
        transpose4(e, l4);
        transpose4(e + 4, h4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo[k] = l4[k]; hi[k] = h4[k]; }
        lo[4] = br[0]; hi[4] = br[1];
    };
    // per-tile lead-in as in dswx_classify_lut: thread i of the tile owns group i - lead, so that lane 0 of every wave
    // stores on a 256-byte boundary of the layers whatever the tile stride (contiguous tiles: 3660 x 3660 = 144 mod 256)
    const uint8_t* const anchor = a.out.wtr ? a.out.wtr : a.out.bwtr ? a.out.bwtr : a.out.conf ? a.out.conf :
                                  a.out.cloud ? a.out.cloud : a.out.browse ? a.out.browse : a.cover_state;
    const int lead = (int)((reinterpret_cast<uintptr_t>(anchor + tile_base) >> 3) & 31u);
    const long long grp0 = (long long)blockIdx.x * FIN_GROUPS * 256 + t - lead;
    if (grp0 >= 0 && (grp0 + (FIN_GROUPS - 1) * 256) * 8 + 8 <= a.n_pixels) {
        // every group of this thread is complete: straight-line code, all loads before the first use
        u32x2 st[FIN_GROUPS];
        uint32_t snow8[FIN_GROUPS];
#pragma unroll
        for (int u = 0; u < FIN_GROUPS; ++u) {
            st[u] = ldg_u<u32x2_u, u32x2, true>(a.cover_state + tile_base + (grp0 + u * 256) * 8);
            snow8[u] = snow_plane[grp0 + u * 256];
        }
#pragma unroll
        for (int u = 0; u < FIN_GROUPS; ++u) {
            const uint32_t s2[2] = {st[u].x, st[u].y};
            uint32_t lo[5], hi[5];
            finish_group(s2, snow8[u], lo, hi);
            const long long off = tile_base + (grp0 + u * 256) * 8;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (planes[k]) stg_u<u32x2_u, u32x2, true>(planes[k] + off, u32x2{lo[k], hi[k]});
        }
        return;
    }
    // the tile's last groups (and the lead-in threads of its first block): byte accesses
    for (int u = 0; u < FIN_GROUPS; ++u) {
        const long long px0 = (grp0 + u * 256) * 8;
        const long long left = a.n_pixels - px0;
        if (left <= 0) break;
        if (px0 < 0) continue;              // the lead-in threads of the tile's first block
        const int n = left >= 8 ? 8 : (int)left;
        const long long off = tile_base + px0;
        uint32_t st[2] = {0u, 0u};
        for (int j = 0; j < n; ++j) st[j >> 2] |= (uint32_t)a.cover_state[off + j] << (8 * (j & 3));
        uint32_t lo[5], hi[5];
        finish_group(st, snow_plane[grp0 + u * 256], lo, hi);
        for (int k = 0; k < 5; ++k) {
            if (!planes[k]) continue;
            for (int j = 0; j < n; ++j) planes[k][off + j] = (uint8_t)((j < 4 ? lo[k] : hi[k]) >> (8 * (j & 3)));
        }
    }
}

// Launches stages 2 and 3 for the `n_tiles` tiles of `c2` (c2.out = the four / five layers they produce).
int dswx_cover_stage2_launch(dswx_ctx* ctx, const KArgs& c2, long long n_tiles, hipStream_t s, char* info,
                             size_t info_len) {
    const int width = c2.width, height = c2.height;
    // ctx->cover_kernel (lab switch): words per window row (4 / 8), + 16 = rows read / written by their own threads
    const int nw = (ctx->cover_kernel & 15) == 4 ? 4 : 8;
    const bool staged = (ctx->cover_kernel & 16) == 0;      // default: staged; + 16 = every thread its own row
    const int tw = 32 * nw - 2 * CP_HALO, th = CP_OUT_H;
    dim3 grid((unsigned)((width + tw - 1) / tw), (unsigned)((height + th - 1) / th), (unsigned)n_tiles);
    if (grid.y > 65535) return dswx_fail(DSWX_ERR_ARG, "raster too tall for one launch");
    if (c2.out.wtr || c2.out.bwtr || c2.out.conf || c2.out.cloud || c2.out.browse) {
        HIP_TRY(hipMemsetAsync(c2.cover_snow, 0, (size_t)n_tiles * (size_t)c2.cover_snow_stride * 4, s));
        if (nw == 4 && staged) hipLaunchKernelGGL((dswx_cover_dilate<4, true>), grid, dim3(256), 0, s, c2);
        else if (nw == 4) hipLaunchKernelGGL((dswx_cover_dilate<4, false>), grid, dim3(256), 0, s, c2);
        else if (staged) hipLaunchKernelGGL((dswx_cover_dilate<8, true>), grid, dim3(256), 0, s, c2);
        else hipLaunchKernelGGL((dswx_cover_dilate<8, false>), grid, dim3(256), 0, s, c2);
        HIP_TRY(hipGetLastError());
        const long long groups = (c2.n_pixels + 7) / 8;
        const long long lead_max = 31;
        dim3 fgrid((unsigned)((groups + lead_max + 256 * FIN_GROUPS - 1) / (256 * FIN_GROUPS)), (unsigned)n_tiles);
        hipLaunchKernelGGL(dswx_cover_finish, fgrid, dim3(256), 0, s, c2);
        HIP_TRY(hipGetLastError());
    }
    snprintf(info, info_len, " + dswx_cover_dilate<%d%s> grid=(%u,%u,%u) + dswx_cover_finish", nw, staged ? "" : ",direct",
             grid.x, grid.y, grid.z);
    return DSWX_OK;
}
