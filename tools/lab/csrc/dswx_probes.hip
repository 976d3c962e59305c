// dswx_probes.hip (libdswx_lab.so, NOT part of the product library) -- roofline calibration kernels behind dswx_stream_probe(): they move the
// fused kernel's bytes in many access shapes with trivial arithmetic (DESIGN.md section 5,
// tools/roofline_probe.py).  Outputs are meaningless by design.
#include <cstring>

#include "dswx_host.h"
#include "dswx_lab.h"

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
        const long long off = (long long)blockIdx.y * a.tile_stride + grp * PPT;
        if (PPT == 8) {
            u32x4 x = {1u, 2u, 3u, (uint32_t)grp};
            u32x2 f = {5u, 6u};
            if (MODE != 2 && MODE != 4) {
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
            if (MODE >= 3) {
                // 16 B per lane from the even lanes: lane 2k covers its own 8 px and lane 2k+1's
                if ((threadIdx.x & 1) == 0) {
                    const u32x4 z = {y.x, y.y, y.x + 7u, y.y + 9u};
                    stg<u32x4, NT>(a.out.wtr1 + off, z);
                    stg<u32x4, NT>(a.out.wtr2 + off, z + 1u);
                    stg<u32x4, NT>(a.out.wtr + off, z + 2u);
                    stg<u32x4, NT>(a.out.bwtr + off, z + 3u);
                    stg<u32x4, NT>(a.out.conf + off, ~z);
                    stg<u32x4, NT>(a.out.cloud + off, z + 5u);
                }
                continue;
            }
            stg<u32x2, NT>(a.out.wtr1 + off, y);
            stg<u32x2, NT>(a.out.wtr2 + off, y + 1u);
            stg<u32x2, NT>(a.out.wtr + off, y + 2u);
            stg<u32x2, NT>(a.out.bwtr + off, y + 3u);
            stg<u32x2, NT>(a.out.conf + off, ~y);
            stg<u32x2, NT>(a.out.cloud + off, y + 5u);
        } else {
            u32x4 x0 = {1u, 2u, 3u, (uint32_t)grp}, x1 = x0, f = x0;
            if (MODE != 2 && MODE != 4) {
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
            if (MODE >= 3) {
                // the wave's 2 KiB of DIAG as two lane-contiguous 1 KiB stores (a real kernel
                // moves the data between lanes first)
                const long long wbase = off - (long long)(threadIdx.x & 63) * 16;       // first pixel of the wave
                stg<u32x4, NT>(a.out.diag + wbase + (threadIdx.x & 63) * 8, x0);
                stg<u32x4, NT>(a.out.diag + wbase + 512 + (threadIdx.x & 63) * 8, x1);
            } else {
                stg<u32x4, NT>(a.out.diag + off, x0);
                stg<u32x4, NT>(a.out.diag + off + 8, x1);
            }
            stg<u32x4, NT>(a.out.wtr1 + off, y);
            stg<u32x4, NT>(a.out.wtr2 + off, y + 1u);
            stg<u32x4, NT>(a.out.wtr + off, y + 2u);
            stg<u32x4, NT>(a.out.bwtr + off, y + 3u);
            stg<u32x4, NT>(a.out.conf + off, ~y);
            stg<u32x4, NT>(a.out.cloud + off, y + 5u);
        }
    }
}

// Four pixels per thread, plain one-group-per-thread shape (round 2): 6 x 8-B + 1 x 4-B loads,
// 1 x 8-B + 6 x 4-B stores -- half the requests in flight per thread of the 8-pixel shape, twice the waves.
template <bool NT>
__global__ __launch_bounds__(256) void dswx_ppt4_probe_k(const KArgs a) {
    const long long grp = (long long)blockIdx.x * 256 + threadIdx.x;
    if (grp >= (a.n_pixels >> 2)) return;
    const long long off = (long long)blockIdx.y * a.tile_stride + grp * 4;
    u32x2 x = ldg<u32x2, NT>(a.in.band[0] + off);
#pragma unroll
    for (int k = 1; k < 6; ++k) x ^= ldg<u32x2, NT>(a.in.band[k] + off);
    const uint32_t f = ldg<uint32_t, NT>(a.in.fmask + off);
    const uint32_t y = x.x ^ x.y ^ f;
    stg<u32x2, NT>(a.out.diag + off, x);
    stg<uint32_t, NT>(a.out.wtr1 + off, y);
    stg<uint32_t, NT>(a.out.wtr2 + off, y + 1u);
    stg<uint32_t, NT>(a.out.wtr + off, y + 2u);
    stg<uint32_t, NT>(a.out.bwtr + off, y + 3u);
    stg<uint32_t, NT>(a.out.conf + off, ~y);
    stg<uint32_t, NT>(a.out.cloud + off, y + 5u);
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
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
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
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
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

// Record layout probe: pixels are grouped in records of CHPX pixels; an input record is
// [6 x int16 plane pieces | fmask piece] = 13*CHPX contiguous bytes and an output
// record [diag | 6 x u8 pieces] = 8*CHPX contiguous bytes.  A block handles 2048 px of
// one record with exactly the fused kernel's per-lane access widths, so all 14 accesses
// of a block fall in two short contiguous spans (record sizes are NOT powers of two).
template <int CHPX, bool NT>
__global__ __launch_bounds__(256) void dswx_record_probe_k(const uint8_t* __restrict__ in_arena,
                                                           uint8_t* __restrict__ out_arena, long long n_blocks) {
    constexpr int SUB = CHPX / 2048;
    const long long b = blockIdx.x;
    if (b >= n_blocks) return;
    const long long rec = b / SUB;
    const int j = (int)(b % SUB);
    const uint8_t* irec = in_arena + rec * (13LL * CHPX);
    uint8_t* orec = out_arena + rec * (8LL * CHPX);
    const int t = threadIdx.x;
    u32x4 x = ldg<u32x4, NT>(irec + j * 4096 + t * 16);
#pragma unroll
    for (int k = 1; k < 6; ++k) x ^= ldg<u32x4, NT>(irec + (long long)k * 2 * CHPX + j * 4096 + t * 16);
    const u32x2 f = ldg<u32x2, NT>(irec + 12LL * CHPX + j * 2048 + t * 8);
    u32x2 y; y.x = x.x ^ x.z ^ f.x; y.y = x.y ^ x.w ^ f.y;
    stg<u32x4, NT>(orec + j * 4096 + t * 16, x);
#pragma unroll
    for (int k = 0; k < 6; ++k) stg<u32x2, NT>(orec + (2LL + k) * CHPX + j * 2048 + t * 8, y + (uint32_t)k);
}

// Cache-policy probe: the fused kernel's access shape through buffer instructions whose
// `aux` bits select the policy (bit 0 = sc0, bit 1 = nt, bit 4 = sc1) separately for loads
// and stores.  One descriptor per plane and tile (built from wave-uniform values).
typedef unsigned int bu32x4 __attribute__((__vector_size__(16)));
typedef unsigned int bu32x2 __attribute__((__vector_size__(8)));
template <int LAUX, int SAUX>
__global__ __launch_bounds__(256) void dswx_policy_probe_k(const KArgs a) {
    const long long tile_off = (long long)blockIdx.y * a.tile_stride;
    const unsigned px = blockIdx.x * 2048u + threadIdx.x * 8u;
    if (px + 8u > (unsigned)a.n_pixels) return;
    const int n = (int)a.n_pixels;
    bu32x4 x = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in.band[k] + tile_off), 0, n * 2, 0x00020000);
        x ^= __builtin_amdgcn_raw_buffer_load_b128(rs, px * 2u, 0, LAUX);
    }
    auto rf = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in.fmask + tile_off), 0, n, 0x00020000);
    const bu32x2 f = __builtin_amdgcn_raw_buffer_load_b64(rf, px, 0, LAUX);
    bu32x2 y = {x[0] ^ x[2] ^ f[0], x[1] ^ x[3] ^ f[1]};
    auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out.diag + tile_off), 0, n * 2, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(x, rd, px * 2u, 0, SAUX);
    uint8_t* const planes[6] = {a.out.wtr1, a.out.wtr2, a.out.wtr, a.out.bwtr, a.out.conf, a.out.cloud};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        auto rp = __builtin_amdgcn_make_buffer_rsrc((void*)(planes[k] + tile_off), 0, n, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(y + (unsigned)k, rp, px, 0, SAUX);
    }
}

// Store-policy probe with GLOBAL instructions: the fused shape, builtin non-temporal loads,
// stores as inline asm so that the sc bits can be combined with nt (the builtins offer
// only plain and nt).  SP: 0 = nt, 1 = sc1 nt, 2 = sc0 sc1 nt, 3 = sc1, 4 = builtin nt (control).
template <int SP>
__device__ __forceinline__ void st16_pol(void* p, u32x4 v) {
    if (SP == 0) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if (SP == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (SP == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (SP == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else stg<u32x4, true>(p, v);
}
template <int SP>
__device__ __forceinline__ void st8_pol(void* p, u32x2 v) {
    if (SP == 0) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if (SP == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (SP == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (SP == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else stg<u32x2, true>(p, v);
}
template <int SP>
__global__ __launch_bounds__(256) void dswx_store_policy_probe_k(const KArgs a) {
    const long long grp = (long long)blockIdx.x * 256 + threadIdx.x;
    if (grp >= (a.n_pixels >> 3)) return;
    const long long off = (long long)blockIdx.y * a.tile_stride + grp * 8;
    u32x4 x = ldg<u32x4, true>(a.in.band[0] + off);
#pragma unroll
    for (int k = 1; k < 6; ++k) x ^= ldg<u32x4, true>(a.in.band[k] + off);
    const u32x2 f = ldg<u32x2, true>(a.in.fmask + off);
    u32x2 y; y.x = x.x ^ x.z ^ f.x; y.y = x.y ^ x.w ^ f.y;
    st16_pol<SP>(a.out.diag + off, x);
    uint8_t* const planes[6] = {a.out.wtr1, a.out.wtr2, a.out.wtr, a.out.bwtr, a.out.conf, a.out.cloud};
#pragma unroll
    for (int k = 0; k < 6; ++k) st8_pol<SP>(planes[k] + off, y + (uint32_t)k);
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
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
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

// THIN warp-specialised data movement: block = 1024 threads (16 waves) over a 2048-px chunk.
// Every wave issues at most TWO 1-KiB LDS-DMA loads (26 pieces over 16 waves) and exactly
// ONE 1-KiB store (16 output pieces), i.e. the per-wave request count of the thin copy probes.
// Phase B folds 2 pixels per thread.  LDS 42 KiB, 2 blocks per CU.
template <bool NT>
__global__ __launch_bounds__(1024) void dswx_ws_thin_probe_k(const KArgs a) {
    constexpr int PX = 2048;
    __shared__ __attribute__((aligned(16))) uint8_t lds_in[6 * PX * 2 + PX];
    __shared__ __attribute__((aligned(16))) uint8_t lds_out[PX * 2 + 6 * PX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
    const long long px0 = (long long)blockIdx.x * PX;
    if (px0 + PX > a.n_pixels) return;       // probe only: whole chunks
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int piece = wave + 16 * h;      // 0..23: plane piece/4, KiB piece%4; 24, 25: fmask halves
        if (piece < 24) {
            const uint8_t* src = reinterpret_cast<const uint8_t*>(a.in.band[piece >> 2]) + (tile_base + px0) * 2 + (piece & 3) * 1024;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + lane * 16), (lptr_t)(lds_in + piece * 1024), 16, 0, NT ? 2 : 0);
        } else if (piece < 26) {
            const uint8_t* src = a.in.fmask + tile_base + px0 + (piece - 24) * 1024;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + lane * 16), (lptr_t)(lds_in + piece * 1024), 16, 0, NT ? 2 : 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // phase B: 2 pixels per thread
    const int t = threadIdx.x;
    uint32_t x = *reinterpret_cast<const uint32_t*>(lds_in + t * 4);
#pragma unroll
    for (int k = 1; k < 6; ++k) x ^= *reinterpret_cast<const uint32_t*>(lds_in + k * (PX * 2) + t * 4);
    const uint32_t f = *reinterpret_cast<const uint16_t*>(lds_in + 6 * (PX * 2) + t * 2);
    *reinterpret_cast<uint32_t*>(lds_out + t * 4) = x;
#pragma unroll
    for (int k = 0; k < 6; ++k) *reinterpret_cast<uint16_t*>(lds_out + PX * 2 + k * PX + t * 2) = (uint16_t)(x ^ f ^ (uint32_t)k);
    __syncthreads();
    // phase C: wave w stores output piece w (4 of DIAG, then 2 per u8 plane)
    uint8_t* const planes[7] = {reinterpret_cast<uint8_t*>(a.out.diag), a.out.wtr1, a.out.wtr2, a.out.wtr,
                                a.out.bwtr, a.out.conf, a.out.cloud};
    if (wave < 4) {
        stg<u32x4, NT>(planes[0] + (tile_base + px0) * 2 + wave * 1024 + lane * 16,
                       *reinterpret_cast<const u32x4*>(lds_out + wave * 1024 + lane * 16));
    } else {
        const int u = (wave - 4) >> 1, sub = (wave - 4) & 1;
        stg<u32x4, NT>(planes[1 + u] + tile_base + px0 + sub * 1024 + lane * 16,
                       *reinterpret_cast<const u32x4*>(lds_out + PX * 2 + u * PX + sub * 1024 + lane * 16));
    }
}

// "Thin-4" access shape: 4 pixels per thread, a wave = 256 pixels, and every wave-level memory
// instruction moves 1 KiB in 16-byte lanes by giving different lane ranges different planes:
//   loads   3 x [lanes 0-31: 512 B of int16 plane 2i | lanes 32-63: 512 B of plane 2i+1]
//           + 1 x [lanes 0-15: 256 B of Fmask]                       -> 3.25 loads per wave
//   stores  1 x [lanes 0-31: DIAG 512 B | 32-47: WTR-1 | 48-63: WTR-2]
//           1 x [0-15: WTR | 16-31: BWTR | 32-47: CONF | 48-63: CLOUD] -> 2 stores per wave
// (the fused kernel has 6.5 loads and 7 stores in flight per wave of 512 pixels).  Probe only:
// the lane <-> pixel redistribution a real kernel would need (ds_bpermute) is not done.
template <bool NT>
__global__ __launch_bounds__(256) void dswx_thin4_probe_k(const KArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile_base = (long long)blockIdx.y * a.tile_stride;
    const long long px0 = (long long)blockIdx.x * 1024 + wave * 256;      // first pixel of this wave
    if (px0 + 256 > a.n_pixels) return;
    const int half = lane >> 5, l32 = lane & 31;
    u32x4 x = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int16_t* plane = half ? a.in.band[2 * i + 1] : a.in.band[2 * i];
        x ^= ldg<u32x4, NT>(reinterpret_cast<const uint8_t*>(plane + tile_base + px0) + l32 * 16);
    }
    if (lane < 16) x ^= ldg<u32x4, NT>(a.in.fmask + tile_base + px0 + lane * 16);
    // stores
    {
        uint8_t* dst;
        if (lane < 32) dst = reinterpret_cast<uint8_t*>(a.out.diag + tile_base + px0) + lane * 16;
        else if (lane < 48) dst = a.out.wtr1 + tile_base + px0 + (lane - 32) * 16;
        else dst = a.out.wtr2 + tile_base + px0 + (lane - 48) * 16;
        stg<u32x4, NT>(dst, x);
        uint8_t* const p4[4] = {a.out.wtr, a.out.bwtr, a.out.conf, a.out.cloud};
        uint8_t* d2 = p4[lane >> 4] + tile_base + px0 + (lane & 15) * 16;
        stg<u32x4, NT>(d2, x + 1u);
    }
}

// Write-shape grid: every wave writes R consecutive 1 KiB pieces (16 B per lane) to each of P
// of the six u8 output planes, plane after plane; the grid covers all six planes.  Tells
// run length per plane (R KiB) from planes-per-wave (P) in the store efficiency.
template <int P, int R, bool NT>
__global__ __launch_bounds__(256) void dswx_write_grid_k(const KArgs a, long long total_px) {
    uint8_t* const planes[6] = {a.out.wtr1, a.out.wtr2, a.out.wtr, a.out.bwtr, a.out.conf, a.out.cloud};
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;     // global wave id
    // a "super-chunk" of R KiB per plane is covered by 6 / P waves (one per group of P planes)
    constexpr int GROUPS = 6 / P;
    const long long chunk = wave / GROUPS;
    const int g = (int)(wave % GROUPS);
    const long long byte0 = chunk * (R * 1024LL);
    if (byte0 + R * 1024LL > total_px) return;
    const u32x4 val = {threadIdx.x, blockIdx.x, 3u, 4u};
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) stg<u32x4, NT>(planes[g * P + p] + byte0 + r * 1024 + lane * 16, val);
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

// Steadily mixed two-stream copy: every block reads 13 x 4 KiB contiguous and writes
// 8 x 4 KiB contiguous (16 B per lane), so that the 13 : 8 read : write ratio of the fused
// kernel holds at every moment of the launch.  (dswx_flat_copy_k above reads one word per
// thread and lets only the first 8/13 of the grid write: its last 5/13 is a pure-read tail.)
template <bool NT>
__global__ __launch_bounds__(256) void dswx_steady_copy_k(const u32x4* __restrict__ src, u32x4* __restrict__ dst,
                                                          long long n_blocks) {
    const long long b = blockIdx.x;
    if (b >= n_blocks) return;
    const u32x4* s = src + b * 13 * 256 + threadIdx.x;
    u32x4* d = dst + b * 8 * 256 + threadIdx.x;
    u32x4 v[13];
#pragma unroll
    for (int j = 0; j < 13; ++j) v[j] = ldg<u32x4, NT>(s + j * 256);
#pragma unroll
    for (int j = 0; j < 8; ++j) stg<u32x4, NT>(d + j * 256, v[j] ^ v[(j + 5) % 13] ^ v[12 - (j & 3)]);
}

// The same steady 13 : 8 mix with ONE word per thread: blocks of 13 waves, every wave loads
// 1 KiB, waves 0..7 also store 1 KiB.  Separates "fat threads" from "traffic mix" as the
// reason why dswx_steady_copy_k is slower than dswx_flat_copy_k.
// L = words per thread: 1 is "thin"; larger L makes the same waves fatter (L loads in flight per
// thread, then L stores) without changing the addresses a block covers.
template <bool NT, int L>
__global__ __launch_bounds__(832) void dswx_steady_copy_small_k(const u32x4* __restrict__ src, u32x4* __restrict__ dst,
                                                                long long n_blocks) {
    const long long b = blockIdx.x;
    if (b >= n_blocks) return;
    u32x4 v[L];
#pragma unroll
    for (int j = 0; j < L; ++j) v[j] = ldg<u32x4, NT>(src + (b * L + j) * 832 + threadIdx.x);
    if (threadIdx.x < 512) {
#pragma unroll
        for (int j = 0; j < L; ++j) stg<u32x4, NT>(dst + (b * L + j) * 512 + threadIdx.x, L == 1 ? v[j] : v[j] ^ v[(j + 1) % L]);
    } else {
        u32x4 x = v[0];
#pragma unroll
        for (int j = 1; j < L; ++j) x ^= v[j];
        if (x.x == 0x9E3779B9u && x.y == 0x7F4A7C15u) dst[0] = x;   // keep the loads alive
    }
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

extern "C" {

int dswx_stream_probe(dswx_ctx_t* ctx, int64_t n_tiles, int64_t n_pixels, int64_t tile_stride,
                      const dswx_planes_in_t* in, const dswx_planes_out_t* out, int variant, void* stream) {
    if (!ctx || !in || !out) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (n_tiles <= 0 || n_tiles > 65535 || n_pixels <= 0 || n_pixels % 16)
        return dswx_fail(DSWX_ERR_ARG, "probe needs 1..65535 tiles of a multiple of 16 pixels");
    if (!out->diag || !out->wtr1 || !out->wtr2 || !out->wtr || !out->bwtr || !out->conf || !out->cloud || !in->fmask)
        return dswx_fail(DSWX_ERR_ARG, "probe needs all seven output planes");
    for (int k = 0; k < 6; ++k)
        if (!in->band[k] || !aligned_to(in->band[k], 16)) return dswx_fail(DSWX_ERR_ALIGN, "band[%d]", k);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    KArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = *in; a.out = *out; a.n_pixels = n_pixels;
    a.tile_stride = tile_stride ? tile_stride : n_pixels;
    // variant = ppt16 | nt << 1 | log2(iters) << 2 | mode << 9 | xcdmap << 11 |
    // block512 << 12 ; bit 8: flat two-stream copy of
    // the same byte counts (needs the planes laid out as DeviceBatch does:
    // band[0..5], fmask contiguous; diag, wtr1.. contiguous)
    if ((variant & 256) && (variant & 1024) && (variant & 4)) {   // same mix, L words per thread: bits 4-5 = log2 L
        const long long total = n_tiles * n_pixels;
        const int lg = (variant >> 4) & 3;
        const long long n_blocks = (total * 13 / 16 / 832) >> lg;
        dim3 grid((unsigned)n_blocks), block(832);
        const u32x4* sp = (const u32x4*)in->band[0];
        u32x4* dp = (u32x4*)out->diag;
#define SMALL(N, LL) hipLaunchKernelGGL((dswx_steady_copy_small_k<N, LL>), grid, block, 0, s, sp, dp, n_blocks)
        if (variant & 2) { if (lg == 0) SMALL(true, 1); else if (lg == 1) SMALL(true, 2); else if (lg == 2) SMALL(true, 4); else SMALL(true, 8); }
        else { if (lg == 0) SMALL(false, 1); else if (lg == 1) SMALL(false, 2); else if (lg == 2) SMALL(false, 4); else SMALL(false, 8); }
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if ((variant & 256) && (variant & 1024)) {   // steadily mixed two-stream copy, bit 1 = nt
        const long long total = n_tiles * n_pixels;
        const long long n_blocks = total / 4096;             // 4096 px = 13 + 8 blocks of 4 KiB
        dim3 grid((unsigned)n_blocks), block(256);
        if (variant & 2) hipLaunchKernelGGL(dswx_steady_copy_k<true>, grid, block, 0, s, (const u32x4*)in->band[0], (u32x4*)out->diag, n_blocks);
        else hipLaunchKernelGGL(dswx_steady_copy_k<false>, grid, block, 0, s, (const u32x4*)in->band[0], (u32x4*)out->diag, n_blocks);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & 256) {
        const long long total = n_tiles * n_pixels;
        const long long n16_in = total * 13 / 16, n16_out = total * 8 / 16;
        dim3 grid((unsigned)((n16_in + 255) / 256)), block(256);
        if (variant & 2) hipLaunchKernelGGL(dswx_flat_copy_k<true>, grid, block, 0, s, (const u32x4*)in->band[0], (u32x4*)out->diag, n16_in, n16_out);
        else hipLaunchKernelGGL(dswx_flat_copy_k<false>, grid, block, 0, s, (const u32x4*)in->band[0], (u32x4*)out->diag, n16_in, n16_out);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & (1 << 26)) {  // four pixels per thread, bit 1 = nt
        dim3 grid((unsigned)(((n_pixels >> 2) + 255) / 256), (unsigned)n_tiles), block(256);
        if (variant & 2) hipLaunchKernelGGL(dswx_ppt4_probe_k<true>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(dswx_ppt4_probe_k<false>, grid, block, 0, s, a);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & 1048576) {  // write-shape grid: bits 2-3 = log2-ish P index {1,2,3,6}, bits 4-5 = log2 R, bit 1 = nt
        const long long total = n_tiles * n_pixels;
        const int pi = (variant >> 2) & 3, ri = (variant >> 4) & 3;
        const bool wnt = variant & 2;
        const int Pv[4] = {1, 2, 3, 6}, Rv[4] = {1, 2, 4, 8};
        const long long waves = (total / (Rv[ri] * 1024LL)) * (6 / Pv[pi]);
        dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define WG(PP, RR) do { if (wnt) hipLaunchKernelGGL((dswx_write_grid_k<PP, RR, true>), grid, block, 0, s, a, total); else hipLaunchKernelGGL((dswx_write_grid_k<PP, RR, false>), grid, block, 0, s, a, total); } while (0)
#define WG_R(PP) do { if (ri == 0) WG(PP, 1); else if (ri == 1) WG(PP, 2); else if (ri == 2) WG(PP, 4); else WG(PP, 8); } while (0)
        if (pi == 0) WG_R(1); else if (pi == 1) WG_R(2); else if (pi == 2) WG_R(3); else WG_R(6);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if ((variant & 524288) && (variant & 8)) {  // thin-4 shape: 4 px per thread, multi-plane 1-KiB instructions
        dim3 grid((unsigned)(n_pixels / 1024), (unsigned)n_tiles), block(256);
        if (variant & 2) hipLaunchKernelGGL(dswx_thin4_probe_k<true>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(dswx_thin4_probe_k<false>, grid, block, 0, s, a);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if ((variant & 524288) && (variant & 4)) {  // thin warp-specialised data movement (16 waves per 2048 px)
        dim3 grid((unsigned)(n_pixels / 2048), (unsigned)n_tiles), block(1024);
        if (variant & 2) hipLaunchKernelGGL(dswx_ws_thin_probe_k<true>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(dswx_ws_thin_probe_k<false>, grid, block, 0, s, a);
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
    if (variant & (1 << 24)) {  // store-policy probe (global instructions): bits 2-4 = SP
        const int sp = (variant >> 2) & 7;
        dim3 grid((unsigned)(((n_pixels >> 3) + 255) / 256), (unsigned)n_tiles), block(256);
        switch (sp) {
        case 0: hipLaunchKernelGGL(dswx_store_policy_probe_k<0>, grid, block, 0, s, a); break;
        case 1: hipLaunchKernelGGL(dswx_store_policy_probe_k<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(dswx_store_policy_probe_k<2>, grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL(dswx_store_policy_probe_k<3>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(dswx_store_policy_probe_k<4>, grid, block, 0, s, a); break;
        }
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & (1 << 23)) {  // cache-policy probe: bits 2-4 = load policy index, bits 5-7 = store policy index
        const int li = (variant >> 2) & 7, si = (variant >> 5) & 7;
        dim3 grid((unsigned)((n_pixels + 2047) / 2048), (unsigned)n_tiles), block(256);
        if (n_pixels * 2 > 2147483647LL) return dswx_fail(DSWX_ERR_ARG, "tile too large for the policy probe");
        // policy table: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt, 6 sc0, 7 sc0 nt
#define POL_S(L, SI) do { switch (SI) { \
        case 0: hipLaunchKernelGGL((dswx_policy_probe_k<L, 0>), grid, block, 0, s, a); break; \
        case 1: hipLaunchKernelGGL((dswx_policy_probe_k<L, 2>), grid, block, 0, s, a); break; \
        case 2: hipLaunchKernelGGL((dswx_policy_probe_k<L, 16>), grid, block, 0, s, a); break; \
        case 3: hipLaunchKernelGGL((dswx_policy_probe_k<L, 17>), grid, block, 0, s, a); break; \
        case 4: hipLaunchKernelGGL((dswx_policy_probe_k<L, 18>), grid, block, 0, s, a); break; \
        case 5: hipLaunchKernelGGL((dswx_policy_probe_k<L, 19>), grid, block, 0, s, a); break; \
        case 6: hipLaunchKernelGGL((dswx_policy_probe_k<L, 1>), grid, block, 0, s, a); break; \
        default: hipLaunchKernelGGL((dswx_policy_probe_k<L, 3>), grid, block, 0, s, a); break; } } while (0)
        switch (li) {
        case 0: POL_S(0, si); break; case 1: POL_S(2, si); break; case 2: POL_S(16, si); break;
        case 3: POL_S(17, si); break; case 4: POL_S(18, si); break; case 5: POL_S(19, si); break;
        case 6: POL_S(1, si); break; default: POL_S(3, si); break;
        }
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (variant & (1 << 22)) {  // record layout: bits 2-3 select CHPX = 2048 << (2*sel), bit 1 = nt
        const long long total = n_tiles * a.tile_stride;
        const int sel = (variant >> 2) & 3;
        const long long n_blocks = total / 2048 / 64 * 64;
        dim3 grid((unsigned)n_blocks), block(256);
        const uint8_t* ia = reinterpret_cast<const uint8_t*>(in->band[0]);
        uint8_t* oa = reinterpret_cast<uint8_t*>(out->diag);
        const bool wnt = variant & 2;
#define REC_LAUNCH(CH) do { if (wnt) hipLaunchKernelGGL((dswx_record_probe_k<CH, true>), grid, block, 0, s, ia, oa, n_blocks); else hipLaunchKernelGGL((dswx_record_probe_k<CH, false>), grid, block, 0, s, ia, oa, n_blocks); } while (0)
        if (sel == 0) REC_LAUNCH(2048); else if (sel == 1) REC_LAUNCH(8192); else if (sel == 2) REC_LAUNCH(32768); else REC_LAUNCH(131072);
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
    const int mode = ((variant >> 9) & 3) + ((variant & (1 << 21)) ? 3 : 0);   // bit 21: +3 (fixed store shapes)
    const bool xcd = variant & 2048, big = variant & 4096;
    const int ppt = ppt16 ? 16 : 8, bs = big ? 512 : 256;
    const int64_t groups = n_pixels / ppt;
    dim3 grid((unsigned)((groups + (int64_t)bs * iters - 1) / ((int64_t)bs * iters)), (unsigned)n_tiles), block(bs);
#define PROBE_LAUNCH(PPT, NT, MODE, XCD, BS) hipLaunchKernelGGL((dswx_stream_probe_k<PPT, NT, MODE, XCD, BS>), grid, block, 0, s, a, iters)
#define PROBE_SEL5(PPT, NT, MODE, XCD) do { if (big) PROBE_LAUNCH(PPT, NT, MODE, XCD, 512); else PROBE_LAUNCH(PPT, NT, MODE, XCD, 256); } while (0)
#define PROBE_SEL4(PPT, NT, MODE) do { if (xcd) PROBE_SEL5(PPT, NT, MODE, true); else PROBE_SEL5(PPT, NT, MODE, false); } while (0)
#define PROBE_SEL3(PPT, NT) do { if (mode == 0) PROBE_SEL4(PPT, NT, 0); else if (mode == 1) PROBE_SEL4(PPT, NT, 1); else if (mode == 2) PROBE_SEL4(PPT, NT, 2); else if (mode == 3) PROBE_SEL4(PPT, NT, 3); else PROBE_SEL4(PPT, NT, 4); } while (0)
#define PROBE_SEL2(PPT) do { if (nt) PROBE_SEL3(PPT, true); else PROBE_SEL3(PPT, false); } while (0)
    if (ppt16) PROBE_SEL2(16); else PROBE_SEL2(8);
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

}  // extern "C"
