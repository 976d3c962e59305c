// dswx_writer.hip -- the raster FORMATS either side of the per-pixel path, on the device (ABI v6).
//
// The reference leaves both to GDAL: reading a band file is inflate + inverse predictor + block -> raster
// (`ReadAsArray`, src/proteus/dswx_hls.py:2136-2302), saving a layer is `save_as_cog` (src/proteus/core.py:7-91):
// NEAREST overviews 4 / 16 / 64 / 128 for the integer layers (:37-46), 512 x 512 blocks, PREDICTOR=2 (integer) or 3
// (floating point), DEFLATE (:60-75); the RGB composites are scale * (float32(band) - offset) with NaN on invalid
// pixels (dswx_hls.py:3013-3036).  Compression stays on the host (libdswx_codec.so); everything between the inflated
// block and the classifier's planes, and between its layers and the block to deflate, is byte shuffling over whole
// rasters -- HBM-bound work with no arithmetic to speak of -- and runs here:
//
//   dswx_untile_v1        inflated blocks (tiles or strips, PREDICTOR 1 / 2) -> row-major plane: one wave per block row,
//                         8 elements per lane, the horizontal accumulation as a wave scan
//   dswx_cog_blocks_v1    plane -> the blocks of the full-resolution image AND of every NEAREST overview level (GDAL's
//                         source-pixel rule), zero-padded edge blocks, horizontal differencing applied: one thread =
//                         8 consecutive elements of a block row = one 8- or 16-byte store
//   dswx_cog_blocks_f32   the same for one Float32 plane with the floating-point predictor (byte planes, MSB first,
//                         byte-differenced over the row; no overviews: the reference's are CUBICSPLINE, host only)
//   dswx_rgb_planes_v1    the three Float32 planes of an RGB / infrared-RGB composite
//
// The kernels work on device memory; the host moves the blocked buffers across PCIe with the copy engine (a kernel
// that gathers single bytes straight from page-locked host memory would multiply the PCIe reads).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>

#include "dswx_host.h"

namespace {

struct CogLevelDev {
    int oh, ow, down, across;
    double ry, rx;                      // source rows / columns per overview row / column (level 0: unused)
    unsigned long long first_group;     // of this level, in units of 8 elements
};

struct CogArgs {
    const void* src;
    void* dst;
    int height, width, tile, predictor, n_levels;
    unsigned long long total_groups;
    CogLevelDev lv[DSWX_COG_MAX_LEVELS];
};

// GDAL's NEAREST overview pick (GDALResampleChunk_Near, as proteus_amd/geotiff.py: overview_nearest restates it):
// src = min(int(0.5 + dst * (N / N_ovr)), N - 1) in double precision, one multiplication and one addition (the build
// has -ffp-contract=off: no fused multiply-add)
__device__ __forceinline__ int near_src(int i, double ratio, int n) {
    const double v = 0.5 + (double)i * ratio;
    const long long k = (long long)v;
    return k < (long long)n - 1 ? (int)k : n - 1;
}

template <typename T>
__global__ __launch_bounds__(256) void dswx_cog_blocks_v1(const CogArgs a) {
    const unsigned long long g = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    if (g >= a.total_groups) return;
    int k = 0;
#pragma unroll
    for (int j = 1; j < DSWX_COG_MAX_LEVELS; ++j)
        if (j < a.n_levels && g >= a.lv[j].first_group) k = j;
    const CogLevelDev L = a.lv[k];
    const unsigned long long idx = g - L.first_group;
    const unsigned gpr = (unsigned)a.tile >> 3;                 // groups per block row
    const unsigned long long per_block = (unsigned long long)gpr * (unsigned)a.tile;
    const unsigned long long b = idx / per_block;
    const unsigned w = (unsigned)(idx % per_block);
    const int y = (int)(w / gpr), xg = (int)(w % gpr);
    const int by = (int)(b / (unsigned)L.across), bx = (int)(b % (unsigned)L.across);
    const int oy = by * a.tile + y, ox0 = bx * a.tile + xg * 8;
    const T* __restrict__ src = static_cast<const T*>(a.src);
    T v[9];                                                     // v[0] = the element left of the group
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = 0;
    if (oy < L.oh) {
        const long long row = (long long)(k == 0 ? oy : near_src(oy, L.ry, a.height)) * a.width;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int ox = ox0 + j - 1;
            if ((j > 0 || xg > 0) && ox < L.ow) v[j] = src[row + (k == 0 ? ox : near_src(ox, L.rx, a.width))];
        }
    }
    T out[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        out[j] = a.predictor == 2 ? (T)(v[j + 1] - v[j]) : v[j + 1];      // (v[0] = 0 at the start of a block row)
    T* dst = static_cast<T*>(a.dst) + g * 8ull;
    if constexpr (sizeof(T) == 1) {
        uint2 pack;
        memcpy(&pack, out, 8);
        *reinterpret_cast<uint2*>(dst) = pack;
    } else {
        uint4 pack;
        memcpy(&pack, out, 16);
        *reinterpret_cast<uint4*>(dst) = pack;
    }
}

struct F32BlockArgs {
    const float* src;
    unsigned char* dst;
    int height, width, tile, down, across;
    unsigned long long total_quads;       // output bytes / 4
};

// TIFF floating-point predictor (Adobe TIFF Technical Note 3, libtiff fpDiff), as geotiff._fp_predictor_encode: per block
// row the bytes of the samples are regrouped into byte planes, most significant first, and the whole row of bytes is
// differenced.  One thread = 4 consecutive output bytes.
__device__ __forceinline__ unsigned f32_row_byte(const float* __restrict__ src, long long row, int x0, int width, bool row_ok,
                                                 int q, int tile) {
    const int plane = q / tile, j = q - plane * tile;           // plane 0 = most significant byte
    const int x = x0 + j;
    if (!row_ok || x >= width) return 0u;
    return (__float_as_uint(src[row + x]) >> (8 * (3 - plane))) & 0xffu;
}

__global__ __launch_bounds__(256) void dswx_cog_blocks_f32(const F32BlockArgs a) {
    const unsigned long long t = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    if (t >= a.total_quads) return;
    const unsigned qpr = (unsigned)a.tile;                      // quads per block row: 4 * tile bytes / 4
    const unsigned long long per_block = (unsigned long long)qpr * (unsigned)a.tile;
    const unsigned long long b = t / per_block;
    const unsigned w = (unsigned)(t % per_block);
    const int y = (int)(w / qpr), q0 = (int)(w % qpr) * 4;
    const int by = (int)(b / (unsigned)a.across), bx = (int)(b % (unsigned)a.across);
    const int oy = by * a.tile + y, x0 = bx * a.tile;
    const bool row_ok = oy < a.height;
    const long long row = (long long)oy * a.width;
    unsigned prev = q0 > 0 ? f32_row_byte(a.src, row, x0, a.width, row_ok, q0 - 1, a.tile) : 0u;
    unsigned pack = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned cur = f32_row_byte(a.src, row, x0, a.width, row_ok, q0 + j, a.tile);
        pack |= ((cur - prev) & 0xffu) << (8 * j);
        prev = cur;
    }
    reinterpret_cast<unsigned*>(a.dst)[t] = pack;
}

struct UntileArgs {
    const void* blocks;
    void* dst;
    int height, width, bw, bh, across, down, predictor;
    long long block_rows;                 // across * down * bh
};

// One wave per block row.  PREDICTOR=2 is a running sum over the row in the sample's own width (libtiff horAcc8 /
// horAcc16: wrap-around), here a per-lane sum of 8 + a wave scan + the carry of the previous 512 elements.
template <typename T>
__global__ __launch_bounds__(256) void dswx_untile_v1(const UntileArgs a) {
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= a.block_rows) return;
    const int lane = threadIdx.x & 63;
    const long long blk = wave / a.bh;
    const int y_in = (int)(wave - blk * a.bh);
    const int by = (int)(blk / a.across), bx = (int)(blk - (long long)by * a.across);
    const int y = by * a.bh + y_in;
    if (y >= a.height) return;
    const T* __restrict__ row = static_cast<const T*>(a.blocks) + (size_t)wave * (size_t)a.bw;
    T* __restrict__ out = static_cast<T*>(a.dst) + (size_t)y * (size_t)a.width;
    const int x_base = bx * a.bw;
    unsigned carry = 0;
    for (int c = 0; c < a.bw; c += 512) {
        const int x_in = c + lane * 8;
        unsigned v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (x_in + j < a.bw) ? (unsigned)row[x_in + j] : 0u;
        if (a.predictor == 2) {
#pragma unroll
            for (int j = 1; j < 8; ++j) v[j] += v[j - 1];
            const unsigned total = v[7];
            unsigned incl = total;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned n = __shfl_up(incl, d, 64);
                if (lane >= d) incl += n;
            }
            const unsigned before = carry + incl - total;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += before;
            carry += __shfl(incl, 63, 64);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = x_base + x_in + j;
            if (x_in + j < a.bw && x < a.width) out[x] = (T)v[j];
        }
    }
}

// Float32 with the floating-point predictor, second step: `acc` = the block rows after the byte-wise running sum (each row
// 4 bw bytes: the four byte planes of its bw samples, most significant first -- libtiff fpAcc) -> the samples of the raster.
struct Fp3Args {
    const unsigned char* acc;
    unsigned* dst;
    int height, width, bw, bh, across;
};

__global__ __launch_bounds__(256) void dswx_untile_fp3_gather(const Fp3Args a) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= a.width) return;
    const int by = y / a.bh, y_in = y - by * a.bh, bx = x / a.bw, j = x - bx * a.bw;
    const unsigned char* __restrict__ row = a.acc + ((size_t)(by * a.across + bx) * (size_t)a.bh + (size_t)y_in) * (size_t)(4 * a.bw);
    const unsigned bits = (unsigned)row[j] << 24 | (unsigned)row[a.bw + j] << 16 | (unsigned)row[2 * a.bw + j] << 8 | (unsigned)row[3 * a.bw + j];
    a.dst[(size_t)y * (size_t)a.width + (size_t)x] = bits;
}

// One separable pass of the CUBICSPLINE overview convolution (geotiff._convolve_axis: GDAL's GDALResampleChunk_Convolution
// restated).  Output element (line r, position j) = sum over taps of src[line r, clamp(first[j] + k)] * w[j][k], normalised over
// the taps that exist and are not NaN; NaN where nothing is left.  The weights come from the host (the same numpy
// expression the host writer uses), the accumulation is float64 in tap order.  Strides in ELEMENTS: one kernel for the
// horizontal pass (lines = rows) and the vertical one (lines = columns).
struct ConvArgs {
    const void* src;
    void* dst;
    const int* first;
    const double* w;
    long long n_lines, n_in, n_out;
    long long src_line_stride, src_elem_stride, dst_line_stride, dst_elem_stride;
    int taps, lines_fastest;            // lines_fastest: threadIdx.x walks the lines (vertical pass: coalesced columns)
};

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dswx_convolve_axis_v1(const ConvArgs a) {
    const long long fast = (long long)blockIdx.x * 256 + threadIdx.x, slow = blockIdx.y;
    const long long r = a.lines_fastest ? fast : slow, j = a.lines_fastest ? slow : fast;
    if (r >= a.n_lines || j >= a.n_out) return;
    const TI* __restrict__ line = static_cast<const TI*>(a.src) + r * a.src_line_stride;
    const double* __restrict__ w = a.w + j;                    // weights [taps][n_out]: neighbouring outputs read neighbouring weights
    const long long f0 = a.first[j];
    double num = 0.0, den = 0.0;
    for (int k = 0; k < a.taps; ++k) {
        long long i = f0 + k;
        i = i < 0 ? 0 : (i >= a.n_in ? a.n_in - 1 : i);
        const double g = (double)line[i * a.src_elem_stride];
        const double ww = (g != g) ? 0.0 : w[k * a.n_out];                // (taps outside the raster carry a zero weight from the host)
        num += (ww > 0.0 ? g : 0.0) * ww;
        den += ww;
    }
    const double out = den > 0.0 ? num / den : __longlong_as_double(0x7ff8000000000000LL);
    static_cast<TO*>(a.dst)[r * a.dst_line_stride + j * a.dst_elem_stride] = (TO)out;
}

// What `gdal_band.WriteArray` stores in a GDT_Byte band (GDALCopyWords; save_dswx_product creates every band of the
// multi-band file as Byte, dswx_hls.py:2663-2666): integers clamped to 0 .. 255, floating point clamped, rounded half up, NaN -> 0.
template <typename T>
__global__ __launch_bounds__(256) void dswx_to_byte_v1(const T* __restrict__ src, unsigned char* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const T v = src[i];
    if constexpr (sizeof(T) == 4) {
        double x = (double)v;
        x = (x != x) ? 0.0 : (x < 0.0 ? 0.0 : (x > 255.0 ? 255.0 : x));
        x = floor(x + 0.5);
        dst[i] = (unsigned char)(x > 255.0 ? 255.0 : x);
    } else {
        dst[i] = (unsigned char)(v < (T)0 ? (T)0 : (v > (T)255 ? (T)255 : v));
    }
}

// dst[i][j] = src[rows[i]][cols[j]]: a nearest-neighbour resampling whose source indices the host computed (the browse PNG of
// geotiff2png: GDAL RasterIO's pick, restated in geotiff.resample_nearest).
template <typename T>
__global__ __launch_bounds__(256) void dswx_gather_2d_v1(const T* __restrict__ src, long long src_width, const int* __restrict__ rows,
                                                        const int* __restrict__ cols, int n_cols, T* __restrict__ dst) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= n_cols) return;
    dst[(long long)i * n_cols + j] = src[(long long)rows[i] * src_width + cols[j]];
}

struct RgbArgs {
    const short* band[3];
    const unsigned short* diag;
    float* out;
    long long n;
    float scale[3], offset[3];
    int clip;
};

// _save_output_rgb_file (dswx_hls.py:3013-3036): scale * (float32(band) - offset) in float32 (a Python float against a
// float32 array is a weak scalar), NaN where the pixel is invalid (DIAG carries the fill code there, :5227).  The bands
// are the CLIPPED reflectances (np.clip(img, 1, None), :2298-2299).
__global__ __launch_bounds__(256) void dswx_rgb_planes_v1(const RgbArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const bool invalid = a.diag && a.diag[i] == 65535;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        short s = a.band[c][i];
        if (a.clip && s < 1) s = 1;
        const float d = (float)s - a.offset[c];
        a.out[(long long)c * a.n + i] = invalid ? __uint_as_float(0x7fc00000u) : a.scale[c] * d;
    }
}

int cog_layout_impl(int64_t height, int64_t width, int32_t elem_bytes, int32_t tile, const int32_t* factors, int32_t n_factors,
                    dswx_cog_layout_t* out) {
    if (!out || (n_factors > 0 && !factors)) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (height <= 0 || width <= 0 || height > 2147483647LL / 2 || width > 2147483647LL / 2)
        return dswx_fail(DSWX_ERR_ARG, "raster size out of range");
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4) return dswx_fail(DSWX_ERR_ARG, "elem_bytes must be 1, 2 or 4");
    if (tile < 8 || tile > 4096 || (tile & 7)) return dswx_fail(DSWX_ERR_ARG, "tile must be a multiple of 8 (16 for TIFF) in 8 .. 4096");
    if (n_factors < 0 || n_factors + 1 > DSWX_COG_MAX_LEVELS) return dswx_fail(DSWX_ERR_ARG, "too many overview levels");
    memset(out, 0, sizeof *out);
    out->tile = tile;
    uint64_t off = 0;
    int n = 0;
    for (int k = -1; k < n_factors; ++k) {
        const int32_t f = k < 0 ? 1 : factors[k];
        if (k >= 0 && f < 1) return dswx_fail(DSWX_ERR_ARG, "overview factor %d", f);
        // write_geotiff's rule: a factor of 1 or a 1 x 1 raster produces no level
        if (k >= 0 && (f == 1 || (height == 1 && width == 1))) continue;
        const int64_t oh = (height + f - 1) / f, ow = (width + f - 1) / f;
        out->factor[n] = f;
        out->height[n] = oh;
        out->width[n] = ow;
        out->blocks_down[n] = (int32_t)((oh + tile - 1) / tile);
        out->blocks_across[n] = (int32_t)((ow + tile - 1) / tile);
        out->offset_bytes[n] = off;
        off += (uint64_t)out->blocks_down[n] * (uint64_t)out->blocks_across[n] * (uint64_t)tile * (uint64_t)tile * (uint64_t)elem_bytes;
        ++n;
    }
    out->n_levels = n;
    out->total_bytes = off;
    return DSWX_OK;
}

}  // namespace

extern "C" {

int dswx_cog_layout(int64_t height, int64_t width, int32_t elem_bytes, int32_t tile, const int32_t* factors, int32_t n_factors,
                    dswx_cog_layout_t* out) {
    return cog_layout_impl(height, width, elem_bytes, tile, factors, n_factors, out);
}

#pragma clang fp contract(off)
int dswx_cog_blocks_device(dswx_ctx_t* ctx, const void* plane, int32_t elem_bytes, int64_t height, int64_t width, int32_t tile,
                           const int32_t* factors, int32_t n_factors, int32_t predictor, void* blocks, void* stream) {
    if (!ctx || !plane || !blocks) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    dswx_cog_layout_t lay;
    if (int rc = cog_layout_impl(height, width, elem_bytes, tile, factors, n_factors, &lay)) return rc;
    if (!aligned_to(blocks, 16) || !aligned_to(plane, (size_t)elem_bytes))
        return dswx_fail(DSWX_ERR_ALIGN, "blocks must be 16-byte aligned, the plane aligned to its samples");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (elem_bytes == 4) {
        if (predictor != 3) return dswx_fail(DSWX_ERR_UNSUPPORTED, "4-byte samples: Float32 with PREDICTOR=3 only");
        if (lay.n_levels != 1)
            return dswx_fail(DSWX_ERR_UNSUPPORTED, "Float32 layers carry CUBICSPLINE overviews (core.py:41-46): build each level with dswx_convolve_axis_device and pass it alone");
        F32BlockArgs a = {};
        a.src = static_cast<const float*>(plane);
        a.dst = static_cast<unsigned char*>(blocks);
        a.height = (int)height; a.width = (int)width; a.tile = tile;
        a.down = lay.blocks_down[0]; a.across = lay.blocks_across[0];
        a.total_quads = lay.total_bytes / 4;
        hipLaunchKernelGGL(dswx_cog_blocks_f32, dim3((unsigned)((a.total_quads + 255) / 256)), dim3(256), 0, s, a);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    if (predictor != 1 && predictor != 2) return dswx_fail(DSWX_ERR_ARG, "integer samples: PREDICTOR 1 or 2");
    CogArgs a = {};
    a.src = plane; a.dst = blocks;
    a.height = (int)height; a.width = (int)width; a.tile = tile; a.predictor = predictor; a.n_levels = lay.n_levels;
    for (int k = 0; k < lay.n_levels; ++k) {
        CogLevelDev& L = a.lv[k];
        L.oh = (int)lay.height[k]; L.ow = (int)lay.width[k];
        L.down = lay.blocks_down[k]; L.across = lay.blocks_across[k];
        L.ry = (double)height / (double)lay.height[k];         // h / oh as Python divides two ints: one correctly rounded division
        L.rx = (double)width / (double)lay.width[k];
        L.first_group = lay.offset_bytes[k] / (uint64_t)elem_bytes / 8;
    }
    a.total_groups = lay.total_bytes / (uint64_t)elem_bytes / 8;
    const dim3 grid((unsigned)((a.total_groups + 255) / 256)), block(256);
    if (a.total_groups > 0xffffffffull * 256ull) return dswx_fail(DSWX_ERR_ARG, "raster too large for one launch");
    if (elem_bytes == 1) hipLaunchKernelGGL(dswx_cog_blocks_v1<unsigned char>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(dswx_cog_blocks_v1<unsigned short>, grid, block, 0, s, a);
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

int dswx_untile_device(dswx_ctx_t* ctx, const void* blocks, int32_t elem_bytes, int64_t height, int64_t width,
                       int32_t block_width, int32_t block_height, int32_t predictor, void* plane, void* stream) {
    if (!ctx || !blocks || !plane) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (height <= 0 || width <= 0 || height > 2147483647LL / 2 || width > 2147483647LL / 2)
        return dswx_fail(DSWX_ERR_ARG, "raster size out of range");
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4) return dswx_fail(DSWX_ERR_ARG, "elem_bytes must be 1, 2 or 4");
    if (block_width < 1 || block_height < 1) return dswx_fail(DSWX_ERR_ARG, "empty block");
    if (predictor != 1 && predictor != 2 && !(predictor == 3 && elem_bytes == 4))
        return dswx_fail(DSWX_ERR_UNSUPPORTED, "PREDICTOR 1 or 2 (any sample size), or 3 with 4-byte samples (Float32)");
    if (!aligned_to(blocks, (size_t)elem_bytes) || !aligned_to(plane, (size_t)elem_bytes))
        return dswx_fail(DSWX_ERR_ALIGN, "buffers must be aligned to their samples");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    UntileArgs a = {};
    a.blocks = blocks; a.dst = plane;
    a.height = (int)height; a.width = (int)width; a.bw = block_width; a.bh = block_height;
    a.across = (int)((width + block_width - 1) / block_width);
    a.down = (int)((height + block_height - 1) / block_height);
    a.predictor = predictor;
    a.block_rows = (long long)a.across * a.down * a.bh;
    if ((long long)block_width * elem_bytes > 2147483647LL / 4) return dswx_fail(DSWX_ERR_ARG, "block too wide");
    if (predictor == 3) {
        // (1) the byte-wise running sum over every block row (4 bw bytes) -- the integer kernel on a "raster" of
        // block_rows x 4 bw bytes that is one block -- into a scratch of the context, (2) the byte planes back into samples
        const size_t need = (size_t)a.block_rows * (size_t)block_width * 4;
        if (need > ctx->untile_bytes) {
            HIP_TRY(hipStreamSynchronize(s));
            if (ctx->untile_tmp) HIP_TRY(hipFree(ctx->untile_tmp));
            ctx->untile_tmp = nullptr; ctx->untile_bytes = 0;
            HIP_TRY(dswx_locked_malloc(&ctx->untile_tmp, need));
            ctx->untile_bytes = need;
        }
        if (a.block_rows > 2147483647LL) return dswx_fail(DSWX_ERR_ARG, "raster too large for one launch");
        UntileArgs b = {};
        b.blocks = blocks; b.dst = ctx->untile_tmp;
        b.height = (int)a.block_rows; b.width = 4 * block_width; b.bw = 4 * block_width; b.bh = (int)a.block_rows;
        b.across = 1; b.down = 1; b.predictor = 2; b.block_rows = a.block_rows;
        const unsigned long long groups = ((unsigned long long)b.block_rows + 3) / 4;
        if (groups > 0x7fffffffull) return dswx_fail(DSWX_ERR_ARG, "raster too large for one launch");
        hipLaunchKernelGGL(dswx_untile_v1<unsigned char>, dim3((unsigned)groups), dim3(256), 0, s, b);
        HIP_TRY(hipGetLastError());
        if (height > 65535) return dswx_fail(DSWX_ERR_ARG, "raster too tall for one launch");
        Fp3Args g = {};
        g.acc = static_cast<const unsigned char*>(ctx->untile_tmp); g.dst = static_cast<unsigned*>(plane);
        g.height = (int)height; g.width = (int)width; g.bw = block_width; g.bh = block_height; g.across = a.across;
        hipLaunchKernelGGL(dswx_untile_fp3_gather, dim3((unsigned)((width + 255) / 256), (unsigned)height), dim3(256), 0, s, g);
        HIP_TRY(hipGetLastError());
        return DSWX_OK;
    }
    const unsigned long long groups = ((unsigned long long)a.block_rows + 3) / 4;
    if (groups > 0x7fffffffull) return dswx_fail(DSWX_ERR_ARG, "raster too large for one launch");
    const dim3 grid((unsigned)groups), block(256);
    if (elem_bytes == 1) hipLaunchKernelGGL(dswx_untile_v1<unsigned char>, grid, block, 0, s, a);
    else if (elem_bytes == 2) hipLaunchKernelGGL(dswx_untile_v1<unsigned short>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(dswx_untile_v1<unsigned int>, grid, block, 0, s, a);
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

int dswx_copy_2d_device(dswx_ctx_t* ctx, void* dst, size_t dst_pitch_bytes, const void* src, size_t src_pitch_bytes,
                        size_t width_bytes, size_t height, void* stream) {
    if (!ctx || !dst || !src) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (width_bytes > dst_pitch_bytes || width_bytes > src_pitch_bytes) return dswx_fail(DSWX_ERR_ARG, "row wider than its pitch");
    if (width_bytes == 0 || height == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch_bytes, src, src_pitch_bytes, width_bytes, height, hipMemcpyDeviceToDevice,
                             stream ? (hipStream_t)stream : ctx->stream));
    return DSWX_OK;
}

int dswx_rgb_planes_device(dswx_ctx_t* ctx, const int16_t* red, const int16_t* green, const int16_t* blue, const uint16_t* diag,
                           int64_t n_pixels, const double scale[3], const double offset[3], int32_t clip_negative_reflectance,
                           float* out, void* stream) {
    if (!ctx || !red || !green || !blue || !scale || !offset || !out) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (n_pixels < 0) return dswx_fail(DSWX_ERR_ARG, "negative size");
    if (n_pixels == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    RgbArgs a = {};
    a.band[0] = red; a.band[1] = green; a.band[2] = blue;
    a.diag = diag; a.out = out; a.n = n_pixels; a.clip = clip_negative_reflectance != 0;
    for (int c = 0; c < 3; ++c) { a.scale[c] = (float)scale[c]; a.offset[c] = (float)offset[c]; }
    const unsigned long long groups = ((unsigned long long)n_pixels + 255) / 256;
    if (groups > 0x7fffffffull) return dswx_fail(DSWX_ERR_ARG, "raster too large for one launch");
    hipLaunchKernelGGL(dswx_rgb_planes_v1, dim3((unsigned)groups), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

int dswx_gather_2d_device(dswx_ctx_t* ctx, const void* src, int32_t elem_bytes, int64_t src_height, int64_t src_width,
                          const int32_t* rows, int32_t n_rows, const int32_t* cols, int32_t n_cols, void* dst, void* stream) {
    if (!ctx || !src || !dst || !rows || !cols) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (src_height < 1 || src_width < 1 || n_rows < 0 || n_cols < 0 || n_rows > 65535)
        return dswx_fail(DSWX_ERR_ARG, "bad size (at most 65535 output rows)");
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4) return dswx_fail(DSWX_ERR_ARG, "elem_bytes must be 1, 2 or 4");
    if (n_rows == 0 || n_cols == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const dim3 grid((unsigned)((n_cols + 255) / 256), (unsigned)n_rows), block(256);
    if (elem_bytes == 1)
        hipLaunchKernelGGL(dswx_gather_2d_v1<unsigned char>, grid, block, 0, s, static_cast<const unsigned char*>(src), (long long)src_width, rows, cols, n_cols, static_cast<unsigned char*>(dst));
    else if (elem_bytes == 2)
        hipLaunchKernelGGL(dswx_gather_2d_v1<unsigned short>, grid, block, 0, s, static_cast<const unsigned short*>(src), (long long)src_width, rows, cols, n_cols, static_cast<unsigned short*>(dst));
    else
        hipLaunchKernelGGL(dswx_gather_2d_v1<unsigned int>, grid, block, 0, s, static_cast<const unsigned int*>(src), (long long)src_width, rows, cols, n_cols, static_cast<unsigned int*>(dst));
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

int dswx_to_byte_device(dswx_ctx_t* ctx, const void* src, int32_t src_kind, int64_t n, uint8_t* dst, void* stream) {
    if (!ctx || !src || !dst) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (n < 0 || src_kind < 1 || src_kind > 3) return dswx_fail(DSWX_ERR_ARG, "src_kind: 1 uint16, 2 int16, 3 float32");
    if (n == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const unsigned long long groups = ((unsigned long long)n + 255) / 256;
    if (groups > 0x7fffffffull) return dswx_fail(DSWX_ERR_ARG, "raster too large for one launch");
    const dim3 grid((unsigned)groups), block(256);
    if (src_kind == 1) hipLaunchKernelGGL(dswx_to_byte_v1<unsigned short>, grid, block, 0, s, static_cast<const unsigned short*>(src), dst, (long long)n);
    else if (src_kind == 2) hipLaunchKernelGGL(dswx_to_byte_v1<short>, grid, block, 0, s, static_cast<const short*>(src), dst, (long long)n);
    else hipLaunchKernelGGL(dswx_to_byte_v1<float>, grid, block, 0, s, static_cast<const float*>(src), dst, (long long)n);
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

int dswx_convolve_axis_device(dswx_ctx_t* ctx, const void* src, int32_t src_is_f64, int64_t n_lines, int64_t n_in,
                              int64_t src_line_stride, int64_t src_elem_stride, int64_t n_out, int32_t taps, const int32_t* first,
                              const double* weights, void* dst, int32_t dst_is_f64, int64_t dst_line_stride,
                              int64_t dst_elem_stride, void* stream) {
    if (!ctx || !src || !dst || !first || !weights) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (n_lines < 0 || n_in < 1 || n_out < 0 || taps < 1 || taps > 4096) return dswx_fail(DSWX_ERR_ARG, "bad size");
    if (n_lines == 0 || n_out == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    ConvArgs a = {};
    a.src = src; a.dst = dst; a.first = first; a.w = weights;
    a.n_lines = n_lines; a.n_in = n_in; a.n_out = n_out; a.taps = taps;
    a.src_line_stride = src_line_stride; a.src_elem_stride = src_elem_stride;
    a.dst_line_stride = dst_line_stride; a.dst_elem_stride = dst_elem_stride;
    // the thread index walks whichever output index is contiguous in memory
    a.lines_fastest = dst_line_stride < dst_elem_stride ? 1 : 0;
    const long long fast = a.lines_fastest ? n_lines : n_out, slow = a.lines_fastest ? n_out : n_lines;
    if (slow > 65535 || (fast + 255) / 256 > 0x7fffffffLL) return dswx_fail(DSWX_ERR_ARG, "raster too large for one launch");
    const dim3 grid((unsigned)((fast + 255) / 256), (unsigned)slow), block(256);
    if (src_is_f64 && dst_is_f64) hipLaunchKernelGGL((dswx_convolve_axis_v1<double, double>), grid, block, 0, s, a);
    else if (src_is_f64) hipLaunchKernelGGL((dswx_convolve_axis_v1<double, float>), grid, block, 0, s, a);
    else if (dst_is_f64) hipLaunchKernelGGL((dswx_convolve_axis_v1<float, double>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((dswx_convolve_axis_v1<float, float>), grid, block, 0, s, a);
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

int dswx_memcpy_h2d_async(dswx_ctx_t* ctx, void* dst, const void* src, size_t bytes, void* stream) {
    if (!ctx) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream ? (hipStream_t)stream : ctx->stream));
    return DSWX_OK;
}

int dswx_memcpy_d2h_async(dswx_ctx_t* ctx, void* dst, const void* src, size_t bytes, void* stream) {
    if (!ctx) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream ? (hipStream_t)stream : ctx->stream));
    return DSWX_OK;
}

}  // extern "C"
