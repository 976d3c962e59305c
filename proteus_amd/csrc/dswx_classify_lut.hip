// dswx_classify_lut.hip -- the production fused kernel on 256-byte aligned batch layouts:
// table-driven compute (dswx_tables.h) with direct register loads and stores.
//
// Block = 256 threads = 2048 consecutive pixels of one tile (grid.y = tile); each thread owns one
// 8-pixel group: 6 x 16-B + 1 x 8-B non-temporal loads (+ 3 x 8 B with LAND / SHAD / OCEAN),
// 1 x 16-B + 6 x 8-B non-temporal stores.
//
// Per-tile lead-in (round 4).  A wave's store to a u8 plane is 512 contiguous bytes; the memory system handles it as
// whole 128-byte lines only if it STARTS on one.  With tile_stride % 256 != 0 -- the reference's natural layout, tiles
// stacked contiguously: 3660 x 3660 = 144 (mod 256) -- tile t starts at pixel t * stride of every plane, so a fixed
// thread -> group mapping puts every wave of every tile but the first across line boundaries on both ends (partial-line
// writes: 3.7 - 4.2 TB/s on the store side instead of 6.3).  The kernel therefore shifts the
// mapping PER TILE: thread i of the tile owns group i - lead, lead = (first pixel of the tile in the plane / 8) mod 32,
// so that lane 0 of every wave sits on a 256-byte boundary of every u8 plane (512 of every int16 plane) whatever the
// stride; the first `lead` threads of a tile idle and the grid carries up to 31 more groups per tile.  It needs all plane
// base pointers 256-byte aligned (so that every plane has the same residue) and tile_stride % 8 == 0; lead is 0 for
// the padded layout.  The tables (2 KiB; 2.5 KiB with masks; + 2 KiB in the EXTRAS
// instantiations) are built on
// the device by dswx_build_tables from the same px_w1 / px_chain / finish_px the scalar kernel
// uses (rebuilt when the parameters change) and copied into LDS by every block.
#include <cstdio>
#include <cstring>
#include <string>

#include "dswx_host.h"
#include "dswx_tables.h"

// WPS: launch bound.  EXTRAS: also the browse plane and the stage-1 scratch of 'cover' mode (the cover
// state byte, looked up in Tables::extra, and the bitmaps of the four dilation predicates).
// The EXTRAS instantiations carry 4.5 KiB of tables per block and, as a straight-line body, 125 VGPRs: as a loop over
// several 8-pixel groups per thread a block reloads a quarter of the tables per pixel and the body fits 90 VGPRs.
// 'cover' mode, 32 tiles, one process, three contexts per build (GB/s of 24 B/px): 1 group 4598 / 4615 / 4610, 2 groups
// 4844 / 4830 / 4825, 4 groups 4898 / 4893 (and 4459 in the context whose scratch planes landed badly: the stage-1
// scratch is placement-sensitive like every other plane).  The plain and masks instantiations
// lose 3 - 8 % with more than one group per thread (5964 -> 5640 -> 5466 GB/s), as in round 1.
constexpr int LUT_EXTRAS_CHUNKS = 4;
constexpr int LUT_DEFAULT_INTERLEAVE = 0;      // see KArgs::tile_interleave and the measurements in DESIGN.md section 5

// FLEX: the per-tile lead-in and the block-order switch.  Off for launches whose every tile starts 256-byte aligned in
// tile-by-tile order (the padded batch layout: the code of rounds 1 - 3, instruction for instruction); with the few scalar
// instructions of FLEX compiled in unconditionally that layout measured 0.1 % (plain) and 1.0 % (LAND / SHAD / OCEAN:
// 107 VGPRs, the tightest instantiation) slower in a same-process A/B against the round-3 build.
// F32: the float32 chain of flag_offset_and_scale_inputs (lut_group<.., F32>): same loads, tables, packing and stores.
template <bool MASKS, bool EXTRAS, int WPS, bool FLEX, bool F32 = false>
__global__ __launch_bounds__(256, WPS) void dswx_classify_lut(const KArgs a, const LutConsts C,
                                                             const Tables* __restrict__ tabs) {
    constexpr int LUT_CHUNKS = EXTRAS ? LUT_EXTRAS_CHUNKS : 1;     // 8-pixel groups per thread
    __shared__ uint32_t s_lut1[128];
    __shared__ uint16_t s_fm16[256];
    __shared__ uint8_t s_land8[MASKS ? 256 : 4];
    __shared__ uint16_t s_pre16[MASKS ? 128 : 2];
    __shared__ uint2 s_chain[128];
    __shared__ uint2 s_extra[EXTRAS ? 256 : 1];
    const DevParams& P = a.P;
    long long n_groups = a.n_pixels >> 3;
    // which tile, which block of it (block-uniform): see KArgs::tile_interleave
    long long tile = blockIdx.y, blk = blockIdx.x;
    if (FLEX && a.tile_interleave > 1) {
        const unsigned G = (unsigned)a.tile_interleave;
        tile = (long long)blockIdx.y * G + blockIdx.x % G;
        blk = blockIdx.x / G;
        if (tile >= a.n_tiles_launch) return;           // the last group of tiles is partial (before any barrier)
    }
    long long tile_base = tile * a.tile_stride;
    if (FLEX && a.ragged) {             // KArgs::ragged: start at the tile's first 8-pixel boundary (block-uniform, SALU)
        const int head = ragged_head(a.in.fmask + tile_base);
        tile_base += head;
        n_groups = a.n_pixels > head ? (a.n_pixels - head) >> 3 : 0;
    }
    {   // 2 KiB of tables (2.5 KiB with masks) per block: one element per thread and table.  (Filling them AFTER the first
        // group's loads have been issued, so that the fill and its barrier overlap the HBM latency, was measured in round 5
        // and LOSES: 1 - 2 % at 1 ... 256 tiles, 9 % on a single tile with masks -- profiles/r05_ab_hoisted_loads.txt.)
        const int i = threadIdx.x;
        if (i < 128) {
            s_lut1[i] = tabs->lut1[i];
            reinterpret_cast<uint32_t*>(s_fm16)[i] = reinterpret_cast<const uint32_t*>(tabs->fm16)[i];
            s_chain[i] = MASKS ? tabs->chainm[i] : tabs->chain[i];
        } else if (MASKS) {
            const int k = i - 128;
            if (k < 64) reinterpret_cast<uint32_t*>(s_land8)[k] = reinterpret_cast<const uint32_t*>(tabs->land8)[k];
            else reinterpret_cast<uint32_t*>(s_pre16)[k - 64] = reinterpret_cast<const uint32_t*>(tabs->pre16)[k - 64];
        }
    }
    if (EXTRAS) s_extra[threadIdx.x] = MASKS ? tabs->extram[threadIdx.x] : tabs->extra[threadIdx.x];
    __syncthreads();

    // groups between the last 256-byte boundary of the u8 planes and this tile's first pixel (wave-uniform, SALU)
    const int lead = FLEX ? (int)((reinterpret_cast<uintptr_t>(a.in.fmask + tile_base) >> 3) & 31u) : 0;
    const bool has_l = MASKS && a.in.land, has_s = MASKS && a.in.shad, has_o = MASKS && a.in.ocean;
    uint32_t cnt = 0, t_ocean = 0;       // cnt: valid in the low half, cloud-and-valid in the high half

    for (int c = 0; c < LUT_CHUNKS; ++c) {
        const long long grp0 = (blk * LUT_CHUNKS + c) * 256 - lead;     // block-uniform
        if (grp0 >= n_groups) break;
        const long long grp = grp0 + threadIdx.x;
        const bool in_range = FLEX ? (unsigned long long)grp < (unsigned long long)n_groups : grp < n_groups;
        // (threads outside the tile redo its first / last group; max(): safe even for a launch without groups)
        const long long off = tile_base + (in_range ? grp : (FLEX && grp < 0 ? 0 : max(n_groups - 1, 0LL))) * 8;
        u32x4 v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = ldg_u<u32x4_u, u32x4, true>(a.in.band[k] + off);
        const u32x2 vf = ldg_u<u32x2_u, u32x2, true>(a.in.fmask + off);
        u32x2 vl = {0u, 0u}, vs = {0x01010101u, 0x01010101u}, vo = {0x01010101u, 0x01010101u};
        if (MASKS) {
            if (has_l) vl = ldg_u<u32x2_u, u32x2, true>(a.in.land + off);
            if (has_s) vs = ldg_u<u32x2_u, u32x2, true>(a.in.shad + off);
            if (has_o) {
                vo = ldg_u<u32x2_u, u32x2, true>(a.in.ocean + off);
                const uint32_t so = __builtin_amdgcn_sad_u8(vo.x, 0u, __builtin_amdgcn_sad_u8(vo.y, 0u, 0u));
                t_ocean += in_range ? so : 0u;
            }
        }
        uint32_t w1w[8], chx[8], chy[8];      // per-pixel table words
        uint32_t idx2[EXTRAS ? 8 : 1];
        lut_group<MASKS, EXTRAS, F32>(P, C, s_lut1, s_fm16, s_land8, s_chain, s_pre16, v, vf, vl, vs, vo, has_l, in_range,
                                 w1w, chx, chy, cnt, idx2);
        if (EXTRAS && in_range) {
            uint32_t ex[8], pa[4], pb[4];     // byte 0 cover state (adjacent bit included), byte 2 browse
            uint32_t bitmaps = 0;             // [snow8, area8, area-and-water8, clear8], pixel j = bit j of every byte
#pragma unroll
            for (int j = 0; j < 8; ++j) { const uint2 e = s_extra[idx2[j]]; ex[j] = e.x; bitmaps |= e.y << j; }
            transpose4(ex, pa); transpose4(ex + 4, pb);
            if (a.cover_state) {              // read back by stages 2 / 3 only after the whole batch: stream them out
                stg_u<u32x2_u, u32x2, true>(a.cover_state + off, u32x2{pa[0], pb[0]});
                __builtin_nontemporal_store(bitmaps, a.cover_bits + tile * a.cover_bits_stride + grp);
            }
            if (a.out.browse) stg_u<u32x2_u, u32x2, true>(a.out.browse + off, u32x2{pa[2], pb[2]});
        }
        if (in_range) {
            GroupPlanes gp;
            lut_pack(w1w, chx, chy, gp);
            if (a.out.diag) stg_u<u32x4_u, u32x4, true>(a.out.diag + off, u32x4{gp.diag[0], gp.diag[1], gp.diag[2], gp.diag[3]});
            if (a.out.wtr1) stg_u<u32x2_u, u32x2, true>(a.out.wtr1 + off, u32x2{gp.w1[0], gp.w1[1]});
            if (a.out.wtr1_aerosol) stg_u<u32x2_u, u32x2, true>(a.out.wtr1_aerosol + off, u32x2{gp.w1a[0], gp.w1a[1]});
            if (a.out.wtr2) stg_u<u32x2_u, u32x2, true>(a.out.wtr2 + off, u32x2{gp.w2[0], gp.w2[1]});
            if (a.out.wtr) stg_u<u32x2_u, u32x2, true>(a.out.wtr + off, u32x2{gp.w[0], gp.w[1]});
            if (a.out.bwtr) stg_u<u32x2_u, u32x2, true>(a.out.bwtr + off, u32x2{gp.bw[0], gp.bw[1]});
            if (a.out.conf) stg_u<u32x2_u, u32x2, true>(a.out.conf + off, u32x2{gp.cf[0], gp.cf[1]});
            if (a.out.cloud) stg_u<u32x2_u, u32x2, true>(a.out.cloud + off, u32x2{gp.cl[0], gp.cl[1]});
        }
    }
    if (a.partials || a.fold_acc) {
        uint32_t c0 = cnt, c2 = t_ocean;
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) { c0 += __shfl_xor(c0, sh); c2 += __shfl_xor(c2, sh); }
        if (a.fold_acc) {
            // Counters folded into this kernel (launches of a few tiles): two levels of last-block-done, ONE atomic per block,
            // no fence.  The blocks of a tile form groups of 2^a.fold_group_log2 blocks (<= 2^17 pixels); every group and every
            // tile has an accumulator on a 128-byte line of its own, so that the groups' atomics spread over the memory
            // channels (one accumulator per tile serialised 6,541 same-address atomics and cost a 3660 x 3660 launch 35 us,
            // and a __threadfence per block -- an L2 write-back on this part -- cost six times the kernel: both measured
            // in round 5).  A block adds  valid | cloud_and_valid << 19 | not_ocean << 38 | ONE TICKET << 57  to its group:
            // the value the atomic returns tells the block whether it drew the group's last ticket, and if so it also holds
            // the group's complete sums.  That block forwards them to the tile: n_not_ocean first (its own word; the atomic
            // RETURNS, so it has been performed), then valid | cloud << 24 | one ticket << 48 -- made data-dependent on that
            // return -- and whoever draws the tile's last ticket reads the n_not_ocean word complete, writes counters[tile]
            // and leaves every accumulator zero for the next launch.  (The host folds only tiles < 2^24 pixels.)
            __shared__ uint2 s_red[4];
            if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = make_uint2(c0, c2);
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned long long v = 0, cl = 0, oc = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { v += s_red[w].x & 0xffffu; cl += s_red[w].x >> 16; oc += s_red[w].y; }
                // (fold_group is a power of two, passed as its log2: a 64-bit division here compiled to ~350 SALU instructions)
                const int sh = a.fold_group_log2;
                const long long gb = 1LL << sh, G = (a.blocks_per_tile + gb - 1) >> sh, g = blk >> sh;
                const long long in_group = min(gb, a.blocks_per_tile - (g << sh));
                unsigned long long* tacc = a.fold_acc + tile * (G + 1) * 16;        // [tile][1 + G] lines of 16 u64
                unsigned long long* gacc = tacc + (1 + g) * 16;
                const unsigned long long add = v | cl << 19 | oc << 38 | 1ull << 57;
                const unsigned long long now = atomicAdd(gacc, add) + add;
                if ((long long)(now >> 57) == in_group) {               // the group's last block: `now` holds the group's sums
                    atomicExch(gacc, 0ull);
                    unsigned long long tadd = (now & 0x7ffffull) | ((now >> 19) & 0x7ffffull) << 24 | 1ull << 48;
                    if (has_o) {
                        unsigned long long seen = atomicAdd(tacc + 1, (now >> 38) & 0x7ffffull);
                        asm volatile("" : "+v"(seen));                  // (opaque: the ticket below waits for this return)
                        tadd += seen >> 63;                             // always 0
                    }
                    const unsigned long long tnow = atomicAdd(tacc, tadd) + tadd;
                    if ((long long)(tnow >> 48) == G) {                 // the tile's last group
                        unsigned long long* out = a.counters + tile * 3;
                        out[0] = tnow & 0xffffffull;
                        out[1] = (tnow >> 24) & 0xffffffull;
                        out[2] = has_o ? atomicExch(tacc + 1, 0ull) : (unsigned long long)(n_groups * 8);
                        atomicExch(tacc, 0ull);
                    }
                }
            }
        } else if ((threadIdx.x & 63) == 0) {
            const long long slot = (tile * a.blocks_per_tile + blk) * 4 + (threadIdx.x >> 6);
            a.partials[slot] = make_uint2(c0, c2);
        }
    }
}

// `lead_max`: the largest per-tile lead-in of the launch (0 when every tile starts 256-byte aligned, else 31 groups)
// log2 of the blocks per group of the folded counters: at most 2^17 pixels per group (19-bit fields of the group accumulator)
static_assert(LUT_EXTRAS_CHUNKS == 4, "dswx_lut_fold_group_log2 assumes 64 / 4 = 16 blocks of 8192 pixels");
int dswx_lut_fold_group_log2(bool extras) { return extras ? 4 : 6; }

void dswx_lut_geometry(const dswx_ctx* ctx, long long groups, bool extras, int lead_max, int* threads, long long* gx) {
    (void)ctx;
    const long long per_block = 256LL * (extras ? LUT_EXTRAS_CHUNKS : 1);
    *threads = 256;
    *gx = (groups + lead_max + per_block - 1) / per_block;
}

int dswx_lut_launch(dswx_ctx* ctx, const KArgs& b, bool masks, dim3 grid, dim3 block, hipStream_t s, char* info,
                    size_t info_len) {
    if (!ctx->tables) HIP_TRY(hipMalloc(&ctx->tables, sizeof(Tables)));
    Tables* tabs = static_cast<Tables*>(ctx->tables);
    LutConsts lc;
    make_lut_consts(b.P, &lc);
    // the tables depend on the parameters only: rebuilt when those change (or the stream does, so
    // that the build always precedes its first use in stream order)
    static_assert(sizeof(DevParams) <= sizeof(ctx->tables_params), "enlarge dswx_ctx::tables_params");
    if (!ctx->tables_valid || ctx->tables_stream != s || std::memcmp(ctx->tables_params, &b.P, sizeof(DevParams)) != 0) {
        hipLaunchKernelGGL(dswx_build_tables, dim3(4), dim3(256), 0, s, b.P, tabs);
        std::memcpy(ctx->tables_params, &b.P, sizeof(DevParams));
        ctx->tables_stream = s;
        ctx->tables_valid = true;
    }
    // launch bound (waves per SIMD), env DSWX_TUNE_LUT_WPS; default 4: 91 VGPRs without masks, 104
    // with LAND / SHAD / OCEAN (5 and 6 spill to scratch there; without masks they give 67 VGPRs
    // and measure 0-2 % slower than 4)
    const int wps = ctx->tune_lut_wps > 0 ? ctx->tune_lut_wps : 4;
    const bool extras = b.out.browse || b.cover_state;
    // block order (KArgs::tile_interleave): `grid` arrives as (blocks per tile, tiles)
    KArgs k = b;
    const long long nt = grid.y;
    long long G = ctx->tune_lut_interleave >= 0 ? ctx->tune_lut_interleave : LUT_DEFAULT_INTERLEAVE;
    if (G > nt) G = nt;
    if (G > 1 && (long long)grid.x * G > 0x7fffffffLL) G = 1;
    k.tile_interleave = (int)(G > 1 ? G : 0);
    k.n_tiles_launch = (int)nt;
    k.blocks_per_tile = grid.x;
    if (G > 1) grid = dim3((unsigned)(grid.x * G), (unsigned)((nt + G - 1) / G));
    // FLEX unless every tile of the launch starts on a 256-byte boundary (then every lead-in is 0) in tile-by-tile order
    const bool flex = G > 1 || (reinterpret_cast<uintptr_t>(b.in.fmask) & 255u) != 0 || (nt > 1 && b.tile_stride % 256 != 0);
#define LUT_LAUNCH(M, E, W) do { if (flex) hipLaunchKernelGGL((dswx_classify_lut<M, E, W, true>), grid, block, 0, s, k, lc, tabs); \
                                 else hipLaunchKernelGGL((dswx_classify_lut<M, E, W, false>), grid, block, 0, s, k, lc, tabs); } while (0)
#define LUT_SEL_W(M, E) do { if (wps >= 6) LUT_LAUNCH(M, E, 6); else if (wps == 5) LUT_LAUNCH(M, E, 5); else LUT_LAUNCH(M, E, 4); } while (0)
    // masks + extras: 125 VGPRs at a bound of 4 (no spill) since the cover bitmaps come from the table; lab A/B: 3
    const bool ex3 = ctx->tune_lut_wps == 3;
    if (b.P.f32_mode) {
        // flag_offset_and_scale_inputs: one instantiation per plane set (launch bound 4, FLEX: the per-tile lead-in is free
        // when every lead is 0)
        if (extras && masks) hipLaunchKernelGGL((dswx_classify_lut<true, true, 4, true, true>), grid, block, 0, s, k, lc, tabs);
        else if (extras) hipLaunchKernelGGL((dswx_classify_lut<false, true, 4, true, true>), grid, block, 0, s, k, lc, tabs);
        else if (masks) hipLaunchKernelGGL((dswx_classify_lut<true, false, 4, true, true>), grid, block, 0, s, k, lc, tabs);
        else hipLaunchKernelGGL((dswx_classify_lut<false, false, 4, true, true>), grid, block, 0, s, k, lc, tabs);
        snprintf(info, info_len, "dswx_classify_lut<%s%s,f32> (table-driven, float32 chain) grid=(%lld,%lld) block=256 wps=4%s%s",
                 masks ? "true" : "false", extras ? ",extras" : "", (long long)k.blocks_per_tile, nt,
                 G > 1 ? (" tiles interleaved x" + std::to_string(G)).c_str() : "", b.fold_acc ? " counters folded" : "");
        return DSWX_OK;
    }
    if (extras) { if (masks && ex3) LUT_LAUNCH(true, true, 3); else if (masks) LUT_LAUNCH(true, true, 4); else LUT_LAUNCH(false, true, 4); }
    else if (masks) LUT_SEL_W(true, false);
    else LUT_SEL_W(false, false);
    snprintf(info, info_len, "dswx_classify_lut<%s%s> (table-driven) grid=(%lld,%lld) block=256 wps=%d%s%s%s",
             masks ? "true" : "false", extras ? ",extras" : "", (long long)k.blocks_per_tile, nt,
             extras ? (masks && ex3 ? 3 : 4) : wps, flex && G <= 1 ? " per-tile lead-in" : "",
             G > 1 ? (" tiles interleaved x" + std::to_string(G)).c_str() : "", b.fold_acc ? " counters folded" : "");
    return DSWX_OK;
}
