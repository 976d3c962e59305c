// dswx_host.h -- host-side internals shared by the translation units of libdswx_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>

#include "dswx_device.h"

struct dswx_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    // grow-only staging for dswx_classify_host
    void* stage = nullptr;
    size_t stage_bytes = 0;
    // grow-only workspace for the vector kernel's per-wave counter partials
    void* partials = nullptr;
    size_t partials_bytes = 0;
    // grow-only accumulators of the folded counters (launches of a few tiles, dswx_classify_lut.hip): zeroed when
    // allocated, left zero by every launch that completes; a launch path that fails marks them dirty
    unsigned long long* fold_acc = nullptr;
    size_t fold_bytes = 0;
    bool fold_clean = false;
    // The workspaces of a context (tables, counter partials, fold accumulators, 'cover' scratch) are shared by all its
    // launches.  Launches on ONE stream are ordered by the stream; a launch on a different stream than the previous one
    // waits for it (dswx_ws_enter / dswx_ws_leave in dswx_hip.hip: an event recorded behind every launch on a caller's
    // stream, or on the context's own stream at the moment the stream changes) -- two launches of one context never
    // overlap on the GPU (the fold accumulators would otherwise be left non-zero for good).
    hipStream_t ws_last = nullptr;         // the stream of the last launch that used the workspaces
    bool ws_used = false;
    hipEvent_t ws_event = nullptr;
    // device copy of the lookup tables of the table-driven kernel (rebuilt per call)
    void* tables = nullptr;
    bool tables_valid = false;             // the device tables match tables_params, built on tables_stream
    hipStream_t tables_stream = nullptr;
    alignas(8) unsigned char tables_params[1024] = {};
    // grow-only scratch of 'cover' mode: state byte per pixel + bitmap dword per 8-pixel group + final snow bits
    void* cover = nullptr;
    size_t cover_bytes = 0;
    // grow-only scratch of dswx_untile_device for Float32 + floating-point predictor (the byte-wise running sums)
    void* untile_tmp = nullptr;
    size_t untile_bytes = 0;
    // pipelined host path (pinned host buffers): copy streams, per-slot events, pinned counter scratch
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
    hipEvent_t pipe_in[3] = {nullptr, nullptr, nullptr}, pipe_k[3] = {nullptr, nullptr, nullptr},
               pipe_out[3] = {nullptr, nullptr, nullptr};
    int64_t* pipe_counters = nullptr;      // hipHostMalloc, [pipe_counters_cap][3]
    size_t pipe_counters_cap = 0;
    // Fixed in production; libdswx_lab.so (experiments, A/B tools, variant tests) changes them through
    // dswx_lab_configure -- the product library reads no environment variable.
    int cover_kernel = 8;                  // 'cover' stage 2: words per window row (8 = 256-column windows, 4 = 128)
    int host_pipeline = 2;                 // page-locked host planes: 2 zero copy, 1 staged three-stream pipeline (lab A/B),
                                           // 0 forces the synchronous path
    int host_chunks = 8;                   // pieces per tile of the pipelined host path
    int shadow_kernel = 0;                 // 2 forces the general one-pixel kernel dswx_shadow_v2 (tests: the exact arithmetic alone);
                                           // 3: the filter kernel in two launches, even block rows then odd (Infinity Cache probe)
    int shadow_grid_pad = 1;               // dswx_shadow_v3: grid.x rounded up to a multiple of this (lab A/B, see the launch)
    std::string last_kernel;
    int tune_lut_wps = 0;    // table-driven kernel: launch bound (4, 5, 6; 0 = automatic)
    int tune_lut_interleave = -1;   // table-driven kernel: tiles whose blocks are interleaved in dispatch order (-1 = the
                                    // product default, dswx_lut_launch: none; 0 / 1 = none).  Measured in round 4 on
                                    // three first-come arenas, 256 tiles, GB/s: none 5976 / 5618 / 6292, G = 4 6006 /
                                    // 5633 / 6397, G = 8 5941 / 5623 / 6354, G = 16 5731 / 5352 / 6285, G = 64 5624 /
                                    // 5201 / 6290, G = 256 5606 / 5214 / 6187 (profiles/r04_ab_interleave.json): spreading
                                    // the 14 streams over many tiles does not lift a badly placed arena and costs 2 - 7 %
                                    // beyond G = 8; kept as a lab switch only
    int tune_wps = 6;        // launch bound of the direct kernel without masks (4, 6, 8)
    int tune_fold = 1;       // 0: always the separate dswx_counters_finish launch (lab A/B of the folded counters)
    int place_force_candidate = -1;   // dswx_batch_place_slide: >= 0 keeps the candidate with that index (packed region at
                                      // offset index * step of the wide range) WHATEVER it measures, without refinement --
                                      // tests only (the kept-placement + trim-while-live case must not hang on a timing)
    int fused_variant = -1;  // -1 automatic = the table-driven kernel (every layout since round 6); 0 forces the direct
                             // kernel, 3 the table-driven one (lab A/B and the variant parity tests)
};

// records a printf-style message for dswx_last_error() and returns `code`
int dswx_fail(int code, const char* fmt, ...);
// validates `p` and derives the kernel parameter block (integer thresholds, exact-quotient
// constants, aerosol table); DSWX_ERR_ARG with a message for unusable thresholds
int dswx_make_dev_params(const dswx_params_t* p, DevParams* d);

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return dswx_fail(DSWX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                             __FILE__, __LINE__);                                                  \
    } while (0)

// The hipMalloc calls of THIS library go through one mutex, the one dswx_batch_pool_trim (dswx_vmm.h) holds while it frees a
// retired range's addresses and reserves them again: between those two calls the addresses are up for grabs, and an
// allocation of another thread of this library must not be the one that takes them (a stress of four threads creating and
// placing batches lost one range in two hundred that way).  The fence covers nothing else: hipHostMalloc, stream and event
// creation, the runtime's own allocations (module load, first launch of a kernel) and every allocation of the caller's
// threads can still land there -- the account reports such a loss (dswx_batch_va_budget: loose_bytes), and the header asks
// callers to trim when nothing else in the process allocates.  (VmRange::destroy itself only retires a range.)
std::mutex& dswx_va_mutex();
template <typename T>
static inline hipError_t dswx_locked_malloc(T** p, size_t n) {
    std::lock_guard<std::mutex> lock(dswx_va_mutex());
    return hipMalloc(reinterpret_cast<void**>(p), n);
}

static inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// Launches of at most this many tiles sum the coverage counters inside the fused kernel (last-block-done) instead of in
// the separate dswx_counters_finish launch, which is ~10 us + a kernel boundary: a quarter of a single-tile call, 0.09 % of
// a 256-tile one (measurements in DESIGN.md section 5).
constexpr int DSWX_FOLD_MAX_TILES = 16;

// ---- table-driven production kernel (dswx_classify_lut.hip)
int dswx_lut_fold_group_log2(bool extras);
void dswx_lut_geometry(const dswx_ctx* ctx, long long groups, bool extras, int lead_max, int* threads, long long* gx);
int dswx_lut_launch(dswx_ctx* ctx, const KArgs& args, bool masks, dim3 grid, dim3 block, hipStream_t stream,
                    char* info, size_t info_len);

// ---- 'cover' mode stage 2 (dswx_cover.hip): appends its description to `info`
int dswx_cover_stage2_launch(dswx_ctx* ctx, const KArgs& c2, long long n_tiles, hipStream_t stream, char* info,
                             size_t info_len);
