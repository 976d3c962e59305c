/*
 * dswx_lab.h -- C ABI of libdswx_lab.so: experiments, calibration probes and A/B switches around
 * the DSWx-HLS classifier.  NOT part of the drop-in boundary (include/dswx_hip.h): nothing here is
 * needed to run the product, and the product library (libdswx_hip.so) carries none of these kernels.
 * libdswx_lab.so links against libdswx_hip.so and works on the same dswx_ctx_t.  Users:
 * tools/roofline_probe.py, tools/ab_variants.py, tests/helpers/fuzz_parity.py, tests/test_gpu_parity.py (both product
 * kernels forced).
 */
#ifndef DSWX_LAB_H
#define DSWX_LAB_H

#include "dswx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* No-op since round 5 (it used to install four experimental kernel structures); kept so that tools written against
 * the earlier lab ABI keep working. */
int dswx_lab_attach(dswx_ctx_t* ctx);

/* A/B switches (what round 1 read from DSWX_* environment variables): "fused_variant" (-1 automatic,
 * 0 direct kernel, 3 table-driven kernel), "tune_wps", "tune_lut_wps", "tune_lut_interleave", "cover_kernel", "host_pipeline", "host_chunks", "shadow_grid_pad", "shadow_kernel" (2: dswx_shadow_v2 always),
 * "place_force_candidate" (>= 0: dswx_batch_place_slide keeps that candidate position whatever it measures; tests). */
int dswx_lab_configure(dswx_ctx_t* ctx, const char* key, int value);

/* Roofline probe: streams exactly the bytes dswx_classify_device streams for the same arguments
 * (7 planes in, 7 planes out, no LAND/SHAD/OCEAN) with trivial arithmetic, to measure the HBM rate
 * that access pattern can reach.  The output planes receive meaningless values.
 * variant = ppt16 | nt << 1 | log2(iters) << 2 | ... (see dswx_probes.hip). */
int dswx_stream_probe(dswx_ctx_t* ctx, int64_t n_tiles, int64_t n_pixels, int64_t tile_stride,
                      const dswx_planes_in_t* in, const dswx_planes_out_t* out, int variant, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSWX_LAB_H */
