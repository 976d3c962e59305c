// dswx_variants.hip -- experimental variants of the fused kernel, selected per context with
// DSWX_FUSED_VARIANT = 1..5.  All are bit-exact (tests/test_gpu_parity.py::
// test_kernel_variants_parity) and all are slower than the default direct-store kernel today;
// they are kept because each isolates one structural idea measured in DESIGN.md section 5.
#include <cstdio>
#include <cstring>
#include <limits>

#include "dswx_host.h"

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
    constexpr int N_CHAIN = MASKS ? 1024 : 128;
    constexpr int IMG = WS_IN_MASKS + (MASKS ? 3 * WS_U8_BYTES : 0);
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * IMG];
    __shared__ uint32_t s_lut1[128];
    __shared__ uint16_t s_fm16[256];
    __shared__ uint8_t s_land8[MASKS ? 256 : 4];
    __shared__ uint2 s_chain[N_CHAIN];
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
    for (int i = threadIdx.x; i < N_CHAIN; i += 256) s_chain[i] = tabs->chain[i];

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
        lut_group<MASKS>(P, C, s_lut1, s_fm16, s_land8, s_chain, v, vf, vl, vs, vo, has_l, in_range, w1w, chx, chy, cnt);
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
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
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

void dswx_variant_geometry(const dswx_ctx* ctx, int variant, long long groups, long long n_tiles, int* threads,
                           long long* gx) {
    const int v = variant;
    *threads = v == 1 ? FUSED_THREADS : 256;
    const int lut_chunks = (ctx->tune_chunks == 1 || ctx->tune_chunks == 4) ? ctx->tune_chunks : 1;
    const long long per_block = (long long)*threads * (v == 3 ? lut_chunks : 1);
    *gx = (groups + per_block - 1) / per_block;
    if (v == 5) {
        // persistent grid: ~tune_pipe_blocks blocks in all, spread evenly over the tiles
        const long long want = (ctx->tune_pipe_blocks + n_tiles - 1) / n_tiles;
        if (want < *gx) *gx = want < 1 ? 1 : want;
    }
}

int dswx_variant_launch(dswx_ctx* ctx, int variant, const KArgs& b, bool masks, dim3 grid, dim3 block, hipStream_t s,
                        char* info, size_t info_len) {
    const long long gx = grid.x, nt = grid.y;
    const bool staged = variant == 1, wspec = variant == 2, tabled = variant == 3, wslut = variant == 4,
               piped = variant == 5;
    const int lut_chunks = (ctx->tune_chunks == 1 || ctx->tune_chunks == 4) ? ctx->tune_chunks : 1;
            if (piped) {
                if (!ctx->tables) HIP_TRY(hipMalloc(&ctx->tables, sizeof(Tables)));
                Tables* tabs = static_cast<Tables*>(ctx->tables);
                LutConsts lc;
                make_lut_consts(b.P, &lc);
                hipLaunchKernelGGL(dswx_build_tables, dim3(4), dim3(256), 0, s, b.P, tabs);
                if (masks) hipLaunchKernelGGL((dswx_classify_pipe<true, 2>), grid, block, 0, s, b, lc, tabs);
                else hipLaunchKernelGGL((dswx_classify_pipe<false, 2>), grid, block, 0, s, b, lc, tabs);
                snprintf(info, info_len, "dswx_classify_pipe<%s> (persistent double-buffered LDS-DMA pipeline) grid=(%lld,%lld) block=256",
                         masks ? "true" : "false", (long long)gx, (long long)nt);
            } else if (wslut) {
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
                snprintf(info, info_len, "dswx_classify_wslut<%s> (warp-specialised + table-driven) grid=(%lld,%lld) block=256 wps=%d",
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
                snprintf(info, info_len, "dswx_classify_lut<%s> (table-driven) grid=(%lld,%lld) block=256 chunks=%d wps=%d",
                         masks ? "true" : "false", (long long)gx, (long long)nt, lut_chunks, wps);
            } else if (wspec) {
                if (masks) hipLaunchKernelGGL(dswx_classify_ws<true>, grid, block, 0, s, b);
                else hipLaunchKernelGGL(dswx_classify_ws<false>, grid, block, 0, s, b);
                snprintf(info, info_len, "dswx_classify_ws<%s> (warp-specialised, LDS-DMA) grid=(%lld,%lld) block=256",
                         masks ? "true" : "false", (long long)gx, (long long)nt);
            } else if (staged) {
                if (masks) hipLaunchKernelGGL(dswx_classify_fused<true>, grid, block, 0, s, b);
                else hipLaunchKernelGGL(dswx_classify_fused<false>, grid, block, 0, s, b);
                snprintf(info, info_len, "dswx_classify_fused<%s> (LDS-staged) grid=(%lld,%lld) block=%d lds=%d",
                         masks ? "true" : "false", (long long)gx, (long long)nt, FUSED_THREADS, STAGE_BYTES);
            }
    return DSWX_OK;
}
