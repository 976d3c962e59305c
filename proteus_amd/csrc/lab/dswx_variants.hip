// dswx_variants.hip (libdswx_lab.so, NOT part of the product library) -- experimental data-movement
// structures of the fused kernel, selected per context with dswx_lab_configure(ctx, "fused_variant",
// 1 | 2 | 4 | 5).  All are bit-exact (tests/test_gpu_parity.py::
// test_kernel_variants_parity) and all are slower than the production kernels (0: dswx_classify_v8
// in dswx_hip.hip, 3: dswx_classify_lut in dswx_classify_lut.hip); they are kept because each
// isolates one structural idea measured in DESIGN.md section 5.
#include <cstdio>
#include <cstring>
#include <limits>

#include "dswx_host.h"
#include "dswx_tables.h"

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
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
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
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
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
    constexpr int N_CHAIN = 128;
    __shared__ __attribute__((aligned(16))) uint8_t lds[WS_IN_MASKS + (MASKS ? 3 * WS_U8_BYTES : 0)];
    __shared__ uint32_t s_lut1[128];
    __shared__ uint16_t s_fm16[256];
    __shared__ uint8_t s_land8[MASKS ? 256 : 4];
    __shared__ uint2 s_chain[N_CHAIN];
    __shared__ uint16_t s_pre16[MASKS ? 128 : 2];
    const DevParams& P = a.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
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
    for (int i = threadIdx.x; i < N_CHAIN; i += 256) s_chain[i] = MASKS ? tabs->chainm[i] : tabs->chain[i];
    if (MASKS) for (int i = threadIdx.x; i < 64; i += 256) reinterpret_cast<uint32_t*>(s_pre16)[i] = reinterpret_cast<const uint32_t*>(tabs->pre16)[i];
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
        lut_group<MASKS>(P, C, s_lut1, s_fm16, s_land8, s_chain, s_pre16, v, vf, vl, vs, vo, has_l, in_range, w1w, chx, chy, cnt);
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

// ------------------------------------------------------------------------------
// Persistent, double-buffered pipeline (DSWX_FUSED_VARIANT=5).  grid = (blocks_per_tile,
// n_tiles); a block of 256 threads walks the 2048-px chunks c = blockIdx.x, +gridDim.x, ...
// of its tile.  Two LDS images X[0], X[1]; iteration i uses X[i & 1] first as the
// input image of chunk i, then (once every thread holds its pixels in registers) as the
// output staging of chunk i:
//
//   wait own LDS-DMA of chunk i (and the global stores of chunk i-1)      s_waitcnt vmcnt(0)
//   barrier B1   image i complete; every LDS read of chunk i-1's store phase has retired
//   issue LDS-DMA of chunk i+1 into X[(i+1) & 1]           <- overlaps everything below
//   registers <- X[i & 1]
//   barrier B2   image i dead
//   lut_group / lut_pack; park plane dwords in X[i & 1]
//   barrier B3
//   store phase: each wave streams consecutive 1 KiB plane pieces from X[i & 1]
//
// The tables are loaded into LDS once per block.  LDS: 2 x 26 KiB (32 with masks) + tables.
// ------------------------------------------------------------------------------
template <bool MASKS>
__device__ __forceinline__ void ws_issue_dma(const KArgs& a, uint8_t* image, long long tile_base, long long px0,
                                             long long last16_i16, long long last16_u8, int wave, int lane,
                                             bool has_l, bool has_s, bool has_o) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int piece = wave * 8 + q;
        if (piece < 24) {
            const int plane = piece >> 2, sub = piece & 3;
            long long byte = (px0 * 2) + sub * 1024 + lane * 16;
            byte = byte <= last16_i16 ? byte : last16_i16;
            __builtin_amdgcn_global_load_lds(
                (gptr_t)(reinterpret_cast<const uint8_t*>(a.in.band[plane]) + tile_base * 2 + byte),
                (lptr_t)(image + plane * WS_BAND_BYTES + sub * 1024), 16, 0, 2);
        } else {
            const int u = (piece - 24) >> 1, sub = (piece - 24) & 1;
            const uint8_t* src = u == 0 ? a.in.fmask : (u == 1 ? a.in.land : (u == 2 ? a.in.shad : a.in.ocean));
            const bool present = u == 0 || (MASKS && ((u == 1 && has_l) || (u == 2 && has_s) || (u == 3 && has_o)));
            if (present) {
                long long byte = px0 + sub * 1024 + lane * 16;
                byte = byte <= last16_u8 ? byte : last16_u8;
                __builtin_amdgcn_global_load_lds((gptr_t)(src + tile_base + byte),
                                                 (lptr_t)(image + WS_IN_FMASK + u * WS_U8_BYTES + sub * 1024), 16, 0, 2);
            }
        }
    }
}

template <bool MASKS, int WPS>
__global__ __launch_bounds__(256, WPS) void dswx_classify_pipe(const KArgs a, const LutConsts C,
                                                              const Tables* __restrict__ tabs) {
    constexpr int N_CHAIN = 128;
    constexpr int IMG = WS_IN_MASKS + (MASKS ? 3 * WS_U8_BYTES : 0);
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * IMG];
    __shared__ uint32_t s_lut1[128];
    __shared__ uint16_t s_fm16[256];
    __shared__ uint8_t s_land8[MASKS ? 256 : 4];
    __shared__ uint2 s_chain[N_CHAIN];
    __shared__ uint16_t s_pre16[MASKS ? 128 : 2];
    const DevParams& P = a.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
    const long long n_groups = a.n_pixels >> 3, n_vec = n_groups << 3;
    const long long n_chunks = (n_vec + WS_PX - 1) / WS_PX;
    const long long last16_i16 = (n_vec - 8) * 2, last16_u8 = n_vec >= 16 ? n_vec - 16 : 0;
    const bool has_l = MASKS && a.in.land, has_s = MASKS && a.in.shad, has_o = MASKS && a.in.ocean;
    const int diag_pieces = a.n_diag_pieces ? 4 : 0;
    const int n_pieces = diag_pieces + 2 * a.n_u8_out;
    const int per_wave = (n_pieces + 3) / 4;

    long long chunk = blockIdx.x;
    if (chunk < n_chunks)
        ws_issue_dma<MASKS>(a, lds, tile_base, chunk * WS_PX, last16_i16, last16_u8, wave, lane, has_l, has_s, has_o);
    for (int i = threadIdx.x; i < 128; i += 256) s_lut1[i] = tabs->lut1[i];
    for (int i = threadIdx.x; i < 128; i += 256) reinterpret_cast<uint32_t*>(s_fm16)[i] = reinterpret_cast<const uint32_t*>(tabs->fm16)[i];
    if (MASKS) for (int i = threadIdx.x; i < 64; i += 256) reinterpret_cast<uint32_t*>(s_land8)[i] = reinterpret_cast<const uint32_t*>(tabs->land8)[i];
    for (int i = threadIdx.x; i < N_CHAIN; i += 256) s_chain[i] = MASKS ? tabs->chainm[i] : tabs->chain[i];
    if (MASKS) for (int i = threadIdx.x; i < 64; i += 256) reinterpret_cast<uint32_t*>(s_pre16)[i] = reinterpret_cast<const uint32_t*>(tabs->pre16)[i];

    uint32_t cnt = 0, t_ocean = 0;
    int buf = 0;
    for (; chunk < n_chunks; chunk += gridDim.x, buf ^= 1) {
        uint8_t* X = lds + buf * IMG;
        const long long px0 = chunk * WS_PX;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                            // B1
        if (chunk + gridDim.x < n_chunks)
            ws_issue_dma<MASKS>(a, lds + (buf ^ 1) * IMG, tile_base, (chunk + gridDim.x) * WS_PX, last16_i16, last16_u8,
                                wave, lane, has_l, has_s, has_o);
        const long long grp = (px0 >> 3) + threadIdx.x;
        const bool in_range = grp < n_groups;
        u32x4 v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = *reinterpret_cast<const u32x4*>(X + k * WS_BAND_BYTES + threadIdx.x * 16);
        const u32x2 vf = *reinterpret_cast<const u32x2*>(X + WS_IN_FMASK + threadIdx.x * 8);
        u32x2 vl = {0u, 0u}, vs = {0x01010101u, 0x01010101u}, vo = {0x01010101u, 0x01010101u};
        if (MASKS) {
            if (has_l) vl = *reinterpret_cast<const u32x2*>(X + WS_IN_MASKS + threadIdx.x * 8);
            if (has_s) vs = *reinterpret_cast<const u32x2*>(X + WS_IN_MASKS + WS_U8_BYTES + threadIdx.x * 8);
            if (has_o) {
                vo = *reinterpret_cast<const u32x2*>(X + WS_IN_MASKS + 2 * WS_U8_BYTES + threadIdx.x * 8);
                const uint32_t so = __builtin_amdgcn_sad_u8(vo.x, 0u, __builtin_amdgcn_sad_u8(vo.y, 0u, 0u));
                t_ocean += in_range ? so : 0u;
            }
        }
        __syncthreads();                                                            // B2
        uint32_t w1w[8], chx[8], chy[8];
        lut_group<MASKS>(P, C, s_lut1, s_fm16, s_land8, s_chain, s_pre16, v, vf, vl, vs, vo, has_l, in_range, w1w, chx, chy, cnt);
        GroupPlanes gp;
        lut_pack(w1w, chx, chy, gp);
        *reinterpret_cast<u32x4*>(X + threadIdx.x * 16) = u32x4{gp.diag[0], gp.diag[1], gp.diag[2], gp.diag[3]};
        uint8_t* su8 = X + WS_OUT_U8 + threadIdx.x * 8;
        *reinterpret_cast<u32x2*>(su8 + 0 * WS_U8_BYTES) = u32x2{gp.w1[0], gp.w1[1]};
        if (a.out.wtr1_aerosol) *reinterpret_cast<u32x2*>(su8 + 1 * WS_U8_BYTES) = u32x2{gp.w1a[0], gp.w1a[1]};
        *reinterpret_cast<u32x2*>(su8 + 2 * WS_U8_BYTES) = u32x2{gp.w2[0], gp.w2[1]};
        *reinterpret_cast<u32x2*>(su8 + 3 * WS_U8_BYTES) = u32x2{gp.w[0], gp.w[1]};
        *reinterpret_cast<u32x2*>(su8 + 4 * WS_U8_BYTES) = u32x2{gp.bw[0], gp.bw[1]};
        *reinterpret_cast<u32x2*>(su8 + 5 * WS_U8_BYTES) = u32x2{gp.cf[0], gp.cf[1]};
        *reinterpret_cast<u32x2*>(su8 + 6 * WS_U8_BYTES) = u32x2{gp.cl[0], gp.cl[1]};
        __syncthreads();                                                            // B3
        for (int q = 0; q < per_wave; ++q) {
            const int piece = wave * per_wave + q;
            if (piece >= n_pieces) break;
            if (piece < diag_pieces) {
                const long long p = px0 + piece * 512 + lane * 8;
                if (p + 8 <= n_vec)
                    stg<u32x4, true>(a.out.diag + tile_base + p, *reinterpret_cast<const u32x4*>(X + piece * 1024 + lane * 16));
            } else {
                const int u = (piece - diag_pieces) >> 1, sub = (piece - diag_pieces) & 1;
                const int region = a.u8_region[u];
                uint8_t* dst = a.u8_out[u] + tile_base;
                const long long p = px0 + sub * 1024 + lane * 16;
                const uint8_t* src = X + WS_OUT_U8 + region * WS_U8_BYTES + sub * 1024 + lane * 16;
                if (p + 16 <= n_vec) stg<u32x4, true>(dst + p, *reinterpret_cast<const u32x4*>(src));
                else if (p + 8 <= n_vec) stg<u32x2, true>(dst + p, *reinterpret_cast<const u32x2*>(src));
            }
        }
    }
    if (a.partials) {
        uint32_t c0 = cnt, c2 = t_ocean;
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) { c0 += __shfl_xor(c0, sh); c2 += __shfl_xor(c2, sh); }
        if (lane == 0) {
            const long long slot = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
            a.partials[slot] = make_uint2(c0, c2);
        }
    }
}




static void dswx_variant_geometry(const dswx_ctx* ctx, int variant, long long groups, long long n_tiles, int* threads,
                           long long* gx) {
    *threads = variant == 1 ? FUSED_THREADS : 256;
    *gx = (groups + *threads - 1) / *threads;
    if (variant == 5) {
        // persistent grid: ~tune_pipe_blocks blocks in all, spread evenly over the tiles
        const long long want = (ctx->lab.tune_pipe_blocks + n_tiles - 1) / n_tiles;
        if (want < *gx) *gx = want < 1 ? 1 : want;
    }
}

static int dswx_variant_launch(dswx_ctx* ctx, int variant, const KArgs& b, bool masks, dim3 grid, dim3 block, hipStream_t s,
                        char* info, size_t info_len) {
    const long long gx = grid.x, nt = grid.y;
    const char* m = masks ? "true" : "false";
    Tables* tabs = nullptr;
    LutConsts lc;
    if (variant == 4 || variant == 5) {
        if (!ctx->tables) HIP_TRY(hipMalloc(&ctx->tables, sizeof(Tables)));
        tabs = static_cast<Tables*>(ctx->tables);
        ctx->tables_valid = false;            // rebuilt below on every call: drop the production kernel's cache
        make_lut_consts(b.P, &lc);
        hipLaunchKernelGGL(dswx_build_tables, dim3(4), dim3(256), 0, s, b.P, tabs);
    }
    switch (variant) {
    case 1:
        if (masks) hipLaunchKernelGGL(dswx_classify_fused<true>, grid, block, 0, s, b);
        else hipLaunchKernelGGL(dswx_classify_fused<false>, grid, block, 0, s, b);
        snprintf(info, info_len, "dswx_classify_fused<%s> (LDS-staged) grid=(%lld,%lld) block=%d lds=%d", m, gx, nt,
                 FUSED_THREADS, STAGE_BYTES);
        break;
    case 2:
        if (masks) hipLaunchKernelGGL(dswx_classify_ws<true>, grid, block, 0, s, b);
        else hipLaunchKernelGGL(dswx_classify_ws<false>, grid, block, 0, s, b);
        snprintf(info, info_len, "dswx_classify_ws<%s> (warp-specialised, LDS-DMA) grid=(%lld,%lld) block=256", m, gx, nt);
        break;
    case 4: {
        const int wps = ctx->tune_lut_wps > 0 ? ctx->tune_lut_wps : 4;
#define WSLUT_LAUNCH(M, W) hipLaunchKernelGGL((dswx_classify_wslut<M, W>), grid, block, 0, s, b, lc, tabs)
        if (masks) { if (wps >= 5) WSLUT_LAUNCH(true, 5); else if (wps == 4) WSLUT_LAUNCH(true, 4); else WSLUT_LAUNCH(true, 3); }
        else if (ctx->lab.tune_ablate == 1) hipLaunchKernelGGL((dswx_classify_wslut<false, 4, 1>), grid, block, 0, s, b, lc, tabs);
        else if (ctx->lab.tune_ablate == 2) hipLaunchKernelGGL((dswx_classify_wslut<false, 4, 2>), grid, block, 0, s, b, lc, tabs);
        else if (ctx->lab.tune_ablate == 3) hipLaunchKernelGGL((dswx_classify_wslut<false, 4, 3>), grid, block, 0, s, b, lc, tabs);
        else { if (wps >= 5) WSLUT_LAUNCH(false, 5); else if (wps == 4) WSLUT_LAUNCH(false, 4); else WSLUT_LAUNCH(false, 3); }
        snprintf(info, info_len, "dswx_classify_wslut<%s> (warp-specialised + table-driven) grid=(%lld,%lld) block=256 wps=%d",
                 m, gx, nt, wps);
        break;
    }
    case 5:
        if (masks) hipLaunchKernelGGL((dswx_classify_pipe<true, 2>), grid, block, 0, s, b, lc, tabs);
        else hipLaunchKernelGGL((dswx_classify_pipe<false, 2>), grid, block, 0, s, b, lc, tabs);
        snprintf(info, info_len, "dswx_classify_pipe<%s> (persistent double-buffered LDS-DMA pipeline) grid=(%lld,%lld) block=256",
                 m, gx, nt);
        break;
    default:
        return dswx_fail(DSWX_ERR_ARG, "unknown kernel variant %d", variant);
    }
    return DSWX_OK;
}


// ==============================================================================
// lab C-ABI (csrc/lab/dswx_lab.h)
// ==============================================================================
#include "dswx_lab.h"

extern "C" {

int dswx_lab_attach(dswx_ctx_t* ctx) {
    if (!ctx) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    ctx->lab.geometry = dswx_variant_geometry;
    ctx->lab.launch = dswx_variant_launch;
    return DSWX_OK;
}

int dswx_lab_configure(dswx_ctx_t* ctx, const char* key, int value) {
    if (!ctx || !key) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    const std::string k = key;
    if (k == "fused_variant") {
        if (value < -1 || value > 5) return dswx_fail(DSWX_ERR_ARG, "fused_variant out of range");
        ctx->lab.fused_variant = value;
    } else if (k == "tune_wps") ctx->tune_wps = value;
    else if (k == "tune_lut_wps") ctx->tune_lut_wps = value;
    else if (k == "tune_lut_interleave") ctx->tune_lut_interleave = value;
    else if (k == "tune_ablate") ctx->lab.tune_ablate = value;
    else if (k == "tune_pipe_blocks") ctx->lab.tune_pipe_blocks = value;
    else if (k == "cover_kernel") ctx->cover_kernel = value;
    else if (k == "host_pipeline") ctx->host_pipeline = value;
    else if (k == "shadow_grid_pad") {
        if (value < 1 || value > 64) return dswx_fail(DSWX_ERR_ARG, "shadow_grid_pad out of range");
        ctx->shadow_grid_pad = value;
    }
    else if (k == "host_chunks") {
        if (value < 1 || value > 256) return dswx_fail(DSWX_ERR_ARG, "host_chunks out of range");
        ctx->host_chunks = value;
    } else return dswx_fail(DSWX_ERR_ARG, "unknown lab key '%s'", key);
    return DSWX_OK;
}

}  // extern "C"
