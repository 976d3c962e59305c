/*
 * dswx_codec.h -- C ABI of libdswx_codec.so: the DEFLATE side of the GeoTIFF / COG reader and writer.
 *
 * What it replaces.  PROTEUS reads and writes every raster through GDAL (C++): `gdal.Open(...).ReadAsArray()`
 * in _load_hls_band_from_file (src/proteus/dswx_hls.py:2136-2302) inflates the blocks of the HLS band files,
 * and `save_as_cog` (src/proteus/core.py:7-91: gdal.Translate with COMPRESS=DEFLATE, 512 x 512 blocks)
 * deflates every block of every product layer.  SURVEY.md section 8 (f4) keeps compression on the host; with
 * the per-pixel chain at milliseconds per tile on the GPU, this codec IS the wall time of a product run
 * (profiles/r06_product_run.json), so it is native, runs the blocks of a file on a pool of threads without
 * Python's interpreter lock, and uses libdeflate when the system has it (as GDAL >= 3.2 builds do), zlib
 * otherwise.  Both produce standard zlib streams (RFC 1950) -- what TIFF compression 8 holds.
 *
 * Host-only: no GPU, no HIP.  Plain pointers and sizes; the caller owns every buffer; int status
 * (0 = ok, negative = error, text from dswx_codec_last_error()).
 */
#ifndef DSWX_CODEC_H
#define DSWX_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSWX_CODEC_ABI_VERSION 1

enum { DSWX_CODEC_OK = 0, DSWX_CODEC_ERR_ARG = -1, DSWX_CODEC_ERR_SPACE = -2, DSWX_CODEC_ERR_DATA = -3 };

int dswx_codec_abi_version(void);
/* "libdeflate" or "zlib": the engine in use (decided once, at the first call). */
const char* dswx_codec_engine(void);
const char* dswx_codec_last_error(void);
/* Tests: on != 0 makes every later call use zlib even when libdeflate is present (both engines are parity-tested). */
int dswx_codec_force_zlib(int on);

/* Processors this process may use: hardware threads, cut down to the container's CPU bandwidth quota (cgroup cpu.max).
 * The pool never runs more workers than this, whatever the calls ask for. */
int dswx_codec_cpu_budget(void);
/* A process that shares the machine with sibling workers (the node-level driver: one worker per GPU) takes only its share:
 * the budget becomes min(processors, what was detected); 0 = back to the detected value. */
int dswx_codec_set_cpu_budget(int processors);

/* Upper bound of the compressed size of `bytes` input bytes (any level, either engine). */
size_t dswx_codec_deflate_bound(size_t bytes);

/* n blocks, block i = src[i] .. src[i] + src_bytes[i], compressed at `level` (1 .. 9; GDAL's DEFLATE default is 6)
 * into dst[i] (capacity dst_cap[i] >= dswx_codec_deflate_bound(src_bytes[i])); dst_bytes[i] receives the size.
 * `threads` workers share the blocks (<= 1: the calling thread alone).  The bytes of a block do not depend on the
 * thread count.  Returns at the first failing block's error. */
int dswx_codec_deflate_blocks(const void* const* src, const size_t* src_bytes, void* const* dst, const size_t* dst_cap,
                              size_t* dst_bytes, int32_t n, int32_t level, int32_t threads);

/* The inverse: n zlib streams into dst[i] (capacity dst_cap[i]); dst_bytes[i] = bytes produced.  A stream that
 * produces MORE than its capacity is an error (DSWX_CODEC_ERR_SPACE); fewer is allowed (a TIFF block may be short). */
int dswx_codec_inflate_blocks(const void* const* src, const size_t* src_bytes, void* const* dst, const size_t* dst_cap,
                              size_t* dst_bytes, int32_t n, int32_t threads);

/* The same for TIFF compression 5 (LZW, TIFF 6.0 section 13, the most-significant-bit-first form libtiff and so GDAL's
 * COMPRESS=LZW write): ancillary rasters (DEM, land-cover maps) handed over by other GDAL tools are often LZW files.  Like
 * libtiff's decoder, a stream stops when its block is full; the pre-6.0 "old-style" variant is refused. */
int dswx_codec_unlzw_blocks(const void* const* src, const size_t* src_bytes, void* const* dst, const size_t* dst_cap,
                            size_t* dst_bytes, int32_t n, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* DSWX_CODEC_H */
