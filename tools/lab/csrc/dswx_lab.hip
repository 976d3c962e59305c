// dswx_lab.hip (libdswx_lab.so, NOT part of the product library) -- the A/B switches of the product library's dispatch.
// The four experimental kernel structures that lived beside them in rounds 1 - 4 (LDS-staged stores, warp-specialised
// LDS-DMA, warp-specialised + table-driven, persistent pipeline: all bit-exact, all slower than the two product
// kernels) were removed in round 5; their measurements are in docs/HISTORY.md and `git show 3b50627:proteus_amd/csrc/lab/dswx_variants.hip`
// has the code.
#include <string>

#include "dswx_host.h"

// ==============================================================================
// lab C-ABI (tools/lab/csrc/dswx_lab.h)
// ==============================================================================
#include "dswx_lab.h"

extern "C" {

int dswx_lab_attach(dswx_ctx_t* ctx) {
    if (!ctx) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    return DSWX_OK;         // nothing to install since round 5 (the four losing kernel structures are gone); kept for callers
}

int dswx_lab_configure(dswx_ctx_t* ctx, const char* key, int value) {
    if (!ctx || !key) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    const std::string k = key;
    if (k == "fused_variant") {
        if (value != -1 && value != 0 && value != 3) return dswx_fail(DSWX_ERR_ARG, "fused_variant: -1 automatic, 0 direct, 3 table-driven");
        ctx->fused_variant = value;
    } else if (k == "tune_wps") ctx->tune_wps = value;
    else if (k == "tune_lut_wps") ctx->tune_lut_wps = value;
    else if (k == "tune_lut_interleave") ctx->tune_lut_interleave = value;
    else if (k == "tune_fold") ctx->tune_fold = value;
    else if (k == "place_force_candidate") ctx->place_force_candidate = value;
    else if (k == "cover_kernel") ctx->cover_kernel = value;
    else if (k == "host_pipeline") ctx->host_pipeline = value;
    else if (k == "shadow_kernel") {
        if (value != 0 && value != 2 && value != 3)
            return dswx_fail(DSWX_ERR_ARG, "shadow_kernel: 0 automatic, 2 the general kernel, 3 the filter kernel in two passes (even / odd block rows)");
        ctx->shadow_kernel = value;
    }
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
