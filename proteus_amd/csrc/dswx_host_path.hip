// dswx_host_path.hip -- dswx_classify_host: the host-pointer entry of the classifier.
// Pageable buffers (and memory a caller page-locked itself): one tile at a time, copy -> classify -> copy on one
// stream.  Buffers from dswx_host_alloc: ZERO COPY -- the kernels read the input planes and write the
// layers across PCIe themselves, both directions at once (3.9 Gpixel/s for 13 B in + 8 B out per pixel,
// measured on four 3660^2 tiles; the staged three-stream pipeline below it reaches 3.05 and stays as a lab A/B).
#include <hip/hip_runtime.h>
#include <cstring>
#include <string>

#include <map>
#include <mutex>

#include "dswx_host.h"

// The page-locked spans this library handed out (dswx_host_alloc): allocated AND resident, the only host memory
// the kernels are let loose on directly (zero copy).  Memory a caller registered itself (hipHostRegister) is
// page-locked in HIP's eyes too, but its pages need not be resident -- see the note in dswx_classify_host.
// (heap objects that are never destroyed: frees that arrive while the process exits must still find them)
static std::mutex& g_spans_mutex = *new std::mutex;
static std::map<uintptr_t, size_t>& g_spans = *new std::map<uintptr_t, size_t>;        // base -> bytes

static bool in_own_span(const void* p, size_t bytes) {
    if (!p) return true;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    std::lock_guard<std::mutex> lock(g_spans_mutex);
    auto it = g_spans.upper_bound(a);
    if (it == g_spans.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
}

extern "C" {

int dswx_host_alloc(dswx_ctx_t* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = bytes ? bytes : 1;
    HIP_TRY(hipHostMalloc(out, n, hipHostMallocDefault));
    std::lock_guard<std::mutex> lock(g_spans_mutex);
    g_spans[reinterpret_cast<uintptr_t>(*out)] = n;
    return DSWX_OK;
}

int dswx_host_free(dswx_ctx_t* ctx, void* ptr) {
    // ctx may be NULL: a span can outlive the context it was allocated through (a host array still alive when its
    // context is destroyed); page-locked host memory is not tied to the current device
    if (!ptr) return DSWX_OK;
    {
        std::lock_guard<std::mutex> lock(g_spans_mutex);
        g_spans.erase(reinterpret_cast<uintptr_t>(ptr));
    }
    if (ctx) HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipHostFree(ptr));
    return DSWX_OK;
}

// Is `p` page-locked host memory HIP knows about (hipHostMalloc / hipHostRegister)?
static bool is_pinned_host(const void* p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();          // plain malloc memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// Zero-copy host path.  Page-locked host memory is mapped into the device's address space, so the
// device-pointer entry can be handed the host planes as they are: the fused kernel's streaming loads become
// PCIe reads, its streaming stores PCIe writes, and the link runs in both directions for the whole launch --
// no staging buffers in HBM, no chunking, no copy engine hand-offs, and every mode (also 'cover': its scratch
// planes stay in HBM, only the inputs and the finished layers cross the link) takes the same path.
static int classify_host_zero_copy(dswx_ctx_t* ctx, const dswx_params_t* params, int64_t n_tiles, int64_t height,
                                   int64_t width, const dswx_planes_in_t* in, const dswx_planes_out_t* out,
                                   int64_t* counters) {
    hipError_t err = hipSuccess;
    auto dev = [&](const void* h) -> void* {
        void* d = nullptr;
        if (h && err == hipSuccess) err = hipHostGetDevicePointer(&d, const_cast<void*>(h), 0);
        return d;
    };
    dswx_planes_in_t din{};
    dswx_planes_out_t dout{};
    for (int k = 0; k < 6; ++k) din.band[k] = static_cast<const int16_t*>(dev(in->band[k]));
    din.fmask = static_cast<const uint8_t*>(dev(in->fmask));
    din.land = static_cast<const uint8_t*>(dev(in->land));
    din.shad = static_cast<const uint8_t*>(dev(in->shad));
    din.ocean = static_cast<const uint8_t*>(dev(in->ocean));
    dout.diag = static_cast<uint16_t*>(dev(out->diag));
    dout.wtr1 = static_cast<uint8_t*>(dev(out->wtr1));
    dout.wtr1_aerosol = static_cast<uint8_t*>(dev(out->wtr1_aerosol));
    dout.wtr2 = static_cast<uint8_t*>(dev(out->wtr2));
    dout.wtr = static_cast<uint8_t*>(dev(out->wtr));
    dout.bwtr = static_cast<uint8_t*>(dev(out->bwtr));
    dout.conf = static_cast<uint8_t*>(dev(out->conf));
    dout.cloud = static_cast<uint8_t*>(dev(out->cloud));
    dout.browse = static_cast<uint8_t*>(dev(out->browse));
    dout.mndwi = static_cast<double*>(dev(out->mndwi));
    dout.ndvi = static_cast<double*>(dev(out->ndvi));
    dout.awesh = static_cast<double*>(dev(out->awesh));
    if (err != hipSuccess) return dswx_fail(DSWX_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(err));
    int64_t* dcnt = nullptr;
    if (counters) {          // the caller's counters may be pageable: a page-locked span of the context receives them
        if ((size_t)n_tiles > ctx->pipe_counters_cap) {
            if (ctx->pipe_counters) HIP_TRY(hipHostFree(ctx->pipe_counters));
            ctx->pipe_counters = nullptr; ctx->pipe_counters_cap = 0;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->pipe_counters), (size_t)n_tiles * 3 * sizeof(int64_t)));
            ctx->pipe_counters_cap = (size_t)n_tiles;
        }
        dcnt = static_cast<int64_t*>(dev(ctx->pipe_counters));
        if (err != hipSuccess) return dswx_fail(DSWX_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(err));
    }
    const int rc = dswx_classify_device_2d(ctx, params, n_tiles, height, width, &din, &dout, dcnt, ctx->stream);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (counters) std::memcpy(counters, ctx->pipe_counters, (size_t)n_tiles * 3 * sizeof(int64_t));
    ctx->last_kernel += " on page-locked host planes (zero copy across PCIe)";
    return DSWX_OK;
}

// Pipelined host path (lab A/B since the zero-copy path: host_pipeline = 1): every tile is cut into `host_chunks` flat pixel ranges (the chain is
// per pixel, so any cut is legal outside 'cover' mode) that flow through three device slots:
// chunk c+1 uploads on the H2D stream while chunk c is classified on the compute stream and
// chunk c-1 downloads on the D2H stream.  Needs page-locked host buffers (dswx_host_alloc or
// hipHostRegister), otherwise the copies are not asynchronous.  Counters are summed per tile
// on the host from per-chunk partial counts.
static int classify_host_pipelined(dswx_ctx_t* ctx, const dswx_params_t* params, int64_t n_tiles, int64_t P,
                                   const dswx_planes_in_t* in, const dswx_planes_out_t* out, int64_t* counters) {
    constexpr int NSLOT = 3;
    if (!ctx->h2d_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->h2d_stream, hipStreamNonBlocking));
    if (!ctx->d2h_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
    for (int i = 0; i < NSLOT; ++i) {
        if (!ctx->pipe_in[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->pipe_in[i], hipEventDisableTiming));
        if (!ctx->pipe_k[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->pipe_k[i], hipEventDisableTiming));
        if (!ctx->pipe_out[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->pipe_out[i], hipEventDisableTiming));
    }
    // chunk size: a multiple of 2048 pixels (one block of the fused kernel), >= 64 Ki pixels
    int64_t chunk = (P + ctx->host_chunks - 1) / ctx->host_chunks;
    if (chunk < 65536) chunk = 65536;
    chunk = (chunk + 2047) / 2048 * 2048;
    const int64_t per_tile = (P + chunk - 1) / chunk;
    const int64_t n_chunks = per_tile * n_tiles;
    if ((size_t)n_chunks > ctx->pipe_counters_cap) {
        if (ctx->pipe_counters) HIP_TRY(hipHostFree(ctx->pipe_counters));
        ctx->pipe_counters = nullptr; ctx->pipe_counters_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->pipe_counters), (size_t)n_chunks * 3 * sizeof(int64_t)));
        ctx->pipe_counters_cap = (size_t)n_chunks;
    }
    // slot layout (offsets within one slot, all 256-byte aligned because chunk % 2048 == 0)
    const uint8_t* const h_in_u8[4] = {in->fmask, in->land, in->shad, in->ocean};
    uint8_t* const h_out_u8[8] = {out->wtr1, out->wtr1_aerosol, out->wtr2, out->wtr, out->bwtr, out->conf, out->cloud,
                                  out->browse};
    double* const h_f64[3] = {out->mndwi, out->ndvi, out->awesh};
    size_t off = 0, o_band[6], o_in_u8[4], o_diag, o_out_u8[8], o_f64[3], o_cnt;
    for (int k = 0; k < 6; ++k) { o_band[k] = off; off += (size_t)chunk * 2; }
    for (int i = 0; i < 4; ++i) { o_in_u8[i] = off; if (h_in_u8[i]) off += (size_t)chunk; }
    o_diag = off; if (out->diag) off += (size_t)chunk * 2;
    for (int i = 0; i < 8; ++i) { o_out_u8[i] = off; if (h_out_u8[i]) off += (size_t)chunk; }
    for (int i = 0; i < 3; ++i) { o_f64[i] = off; if (h_f64[i]) off += (size_t)chunk * 8; }
    o_cnt = off; off += 256;
    const size_t slot_bytes = off;
    if (slot_bytes * NSLOT > ctx->stage_bytes) {
        HIP_TRY(hipDeviceSynchronize());
        if (ctx->stage) HIP_TRY(hipFree(ctx->stage));
        ctx->stage = nullptr; ctx->stage_bytes = 0;
        HIP_TRY(dswx_locked_malloc(&ctx->stage, slot_bytes * NSLOT));
        ctx->stage_bytes = slot_bytes * NSLOT;
    }
    char* const arena = static_cast<char*>(ctx->stage);
    hipStream_t sc = ctx->stream, sh = ctx->h2d_stream, sd = ctx->d2h_stream;
    bool slot_used[NSLOT] = {false, false, false};
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int slot = (int)(c % NSLOT);
        const int64_t tile = c / per_tile, px0 = (c % per_tile) * chunk;
        const int64_t n = (P - px0 < chunk) ? P - px0 : chunk;
        const size_t hoff = (size_t)tile * (size_t)P + (size_t)px0;
        char* base = arena + slot_bytes * slot;
        dswx_planes_in_t din{};
        dswx_planes_out_t dout{};
        // ---- upload (after the slot's previous download finished)
        if (slot_used[slot]) HIP_TRY(hipStreamWaitEvent(sh, ctx->pipe_out[slot], 0));
        for (int k = 0; k < 6; ++k) {
            HIP_TRY(hipMemcpyAsync(base + o_band[k], in->band[k] + hoff, (size_t)n * 2, hipMemcpyHostToDevice, sh));
            din.band[k] = reinterpret_cast<const int16_t*>(base + o_band[k]);
        }
        const uint8_t** const d_in_u8[4] = {&din.fmask, &din.land, &din.shad, &din.ocean};
        for (int i = 0; i < 4; ++i)
            if (h_in_u8[i]) {
                HIP_TRY(hipMemcpyAsync(base + o_in_u8[i], h_in_u8[i] + hoff, (size_t)n, hipMemcpyHostToDevice, sh));
                *d_in_u8[i] = reinterpret_cast<const uint8_t*>(base + o_in_u8[i]);
            }
        HIP_TRY(hipEventRecord(ctx->pipe_in[slot], sh));
        // ---- classify
        if (out->diag) dout.diag = reinterpret_cast<uint16_t*>(base + o_diag);
        uint8_t** const d_out_u8[8] = {&dout.wtr1, &dout.wtr1_aerosol, &dout.wtr2, &dout.wtr, &dout.bwtr, &dout.conf,
                                       &dout.cloud, &dout.browse};
        for (int i = 0; i < 8; ++i) if (h_out_u8[i]) *d_out_u8[i] = reinterpret_cast<uint8_t*>(base + o_out_u8[i]);
        double** const d_f64[3] = {&dout.mndwi, &dout.ndvi, &dout.awesh};
        for (int i = 0; i < 3; ++i) if (h_f64[i]) *d_f64[i] = reinterpret_cast<double*>(base + o_f64[i]);
        int64_t* dcnt = counters ? reinterpret_cast<int64_t*>(base + o_cnt) : nullptr;
        HIP_TRY(hipStreamWaitEvent(sc, ctx->pipe_in[slot], 0));
        const int rc = dswx_classify_device(ctx, params, 1, n, &din, &dout, dcnt, sc);
        if (rc) { (void)hipDeviceSynchronize(); return rc; }
        HIP_TRY(hipEventRecord(ctx->pipe_k[slot], sc));
        // ---- download
        HIP_TRY(hipStreamWaitEvent(sd, ctx->pipe_k[slot], 0));
        if (out->diag) HIP_TRY(hipMemcpyAsync(out->diag + hoff, dout.diag, (size_t)n * 2, hipMemcpyDeviceToHost, sd));
        for (int i = 0; i < 8; ++i)
            if (h_out_u8[i]) HIP_TRY(hipMemcpyAsync(h_out_u8[i] + hoff, *d_out_u8[i], (size_t)n, hipMemcpyDeviceToHost, sd));
        for (int i = 0; i < 3; ++i)
            if (h_f64[i]) HIP_TRY(hipMemcpyAsync(h_f64[i] + hoff, *d_f64[i], (size_t)n * 8, hipMemcpyDeviceToHost, sd));
        if (counters)
            HIP_TRY(hipMemcpyAsync(ctx->pipe_counters + c * 3, dcnt, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, sd));
        HIP_TRY(hipEventRecord(ctx->pipe_out[slot], sd));
        slot_used[slot] = true;
    }
    HIP_TRY(hipStreamSynchronize(sd));
    HIP_TRY(hipStreamSynchronize(sc));
    HIP_TRY(hipStreamSynchronize(sh));
    if (counters)
        for (int64_t t = 0; t < n_tiles; ++t)
            for (int j = 0; j < 3; ++j) {
                int64_t sum = 0;
                for (int64_t q = 0; q < per_tile; ++q) sum += ctx->pipe_counters[(t * per_tile + q) * 3 + j];
                counters[t * 3 + j] = sum;
            }
    ctx->last_kernel += " x" + std::to_string(n_chunks) + " chunks, pipelined over 3 streams (pinned host buffers)";
    return DSWX_OK;
}

int dswx_classify_host(dswx_ctx_t* ctx, const dswx_params_t* params, int64_t n_tiles, int64_t height,
                       int64_t width, const dswx_planes_in_t* in, const dswx_planes_out_t* out,
                       int64_t* counters) {
    if (!ctx || !params || !in || !out) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (n_tiles < 0 || height < 0 || width < 0) return dswx_fail(DSWX_ERR_ARG, "negative size");
    for (int k = 0; k < 6; ++k)
        if (!in->band[k]) return dswx_fail(DSWX_ERR_ARG, "band[%d] is NULL", k);
    if (!in->fmask) return dswx_fail(DSWX_ERR_ARG, "fmask is NULL");
    {   // validate parameters before touching the device
        DevParams tmp;
        int rc = dswx_make_dev_params(params, &tmp);
        if (rc) return rc;
    }
    const int64_t P = height * width;
    if (n_tiles == 0 || P == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->host_pipeline == 2 || (ctx->host_pipeline && params->mask_adjacent_to_cloud_mode != DSWX_ADJ_COVER)) {
        // zero copy (product): every plane inside a span of dswx_host_alloc; staged pipeline (lab): any page-locked
        // memory, first AND last byte of every plane
        const size_t npx = (size_t)n_tiles * (size_t)P;
        const bool own = ctx->host_pipeline == 2;
        auto ok_span = [&](const void* p, size_t bytes) {
            if (!p) return true;
            if (own) return in_own_span(p, bytes);
            return is_pinned_host(p) && is_pinned_host(static_cast<const char*>(p) + bytes - 1);
        };
        bool pinned = true;
        for (int k = 0; k < 6 && pinned; ++k) pinned = ok_span(in->band[k], npx * 2);
        const void* const rest[] = {in->fmask, in->land, in->shad, in->ocean, out->wtr1, out->wtr1_aerosol,
                                    out->wtr2, out->wtr, out->bwtr, out->conf, out->cloud, out->browse};
        for (const void* p : rest) pinned = pinned && ok_span(p, npx);
        pinned = pinned && ok_span(out->diag, npx * 2) && ok_span(out->mndwi, npx * 8) &&
                 ok_span(out->ndvi, npx * 8) && ok_span(out->awesh, npx * 8);
        if (pinned && ctx->host_pipeline == 2) return classify_host_zero_copy(ctx, params, n_tiles, height, width, in, out, counters);
        if (pinned) return classify_host_pipelined(ctx, params, n_tiles, P, in, out, counters);
    }
    // (Page-locking pageable planes IN PLACE for the duration of the call -- hipHostRegister on page-merged ranges, then
    // the zero-copy path -- was built and measured in round 2: 4.5 - 5.2 ms instead of 10.7 ms per 3660^2 tile from plain
    // numpy arrays.  It is NOT used: about one run in twenty of the parity suite came back with a few wrong pixels in a
    // freshly allocated, never-touched output array.  Transiently registered user memory is not pinned the way
    // hipHostMalloc memory is (the driver follows the CPU page tables through MMU notifiers), and kernel WRITES into
    // pages that are being faulted in, migrated or collapsed underneath were not reliable.  Pageable planes are copied.)
    // pageable host buffers (or 'cover' mode): one tile at a time through a grow-only device
    // arena: planes at 256-byte aligned offsets so the vector kernel is always eligible
    auto rnd = [](size_t x) { return (x + 255) & ~size_t(255); };
    size_t off = 0;
    size_t o_band[6], o_fm, o_land = 0, o_shad = 0, o_ocean = 0;
    for (int k = 0; k < 6; ++k) { o_band[k] = off; off += rnd((size_t)P * 2); }
    o_fm = off; off += rnd((size_t)P);
    if (in->land) { o_land = off; off += rnd((size_t)P); }
    if (in->shad) { o_shad = off; off += rnd((size_t)P); }
    if (in->ocean) { o_ocean = off; off += rnd((size_t)P); }
    size_t o_diag = off; if (out->diag) off += rnd((size_t)P * 2);
    uint8_t* const h_u8[8] = {out->wtr1, out->wtr1_aerosol, out->wtr2, out->wtr, out->bwtr, out->conf, out->cloud,
                              out->browse};
    size_t o_u8[8];
    for (int i = 0; i < 8; ++i) { o_u8[i] = off; if (h_u8[i]) off += rnd((size_t)P); }
    double* const h_f64[3] = {out->mndwi, out->ndvi, out->awesh};
    size_t o_f64[3];
    for (int i = 0; i < 3; ++i) { o_f64[i] = off; if (h_f64[i]) off += rnd((size_t)P * 8); }
    size_t o_cnt = off; off += 256;
    if (off > ctx->stage_bytes) {
        if (ctx->stage) HIP_TRY(hipFree(ctx->stage));
        ctx->stage = nullptr; ctx->stage_bytes = 0;
        HIP_TRY(dswx_locked_malloc(&ctx->stage, off));
        ctx->stage_bytes = off;
    }
    char* base = static_cast<char*>(ctx->stage);
    hipStream_t s = ctx->stream;
    for (int64_t t = 0; t < n_tiles; ++t) {
        const size_t sh = (size_t)t * (size_t)P;
        dswx_planes_in_t din{};
        dswx_planes_out_t dout{};
        for (int k = 0; k < 6; ++k) {
            HIP_TRY(hipMemcpyAsync(base + o_band[k], in->band[k] + sh, (size_t)P * 2, hipMemcpyHostToDevice, s));
            din.band[k] = reinterpret_cast<const int16_t*>(base + o_band[k]);
        }
        HIP_TRY(hipMemcpyAsync(base + o_fm, in->fmask + sh, (size_t)P, hipMemcpyHostToDevice, s));
        din.fmask = reinterpret_cast<const uint8_t*>(base + o_fm);
        if (in->land) { HIP_TRY(hipMemcpyAsync(base + o_land, in->land + sh, (size_t)P, hipMemcpyHostToDevice, s)); din.land = reinterpret_cast<const uint8_t*>(base + o_land); }
        if (in->shad) { HIP_TRY(hipMemcpyAsync(base + o_shad, in->shad + sh, (size_t)P, hipMemcpyHostToDevice, s)); din.shad = reinterpret_cast<const uint8_t*>(base + o_shad); }
        if (in->ocean) { HIP_TRY(hipMemcpyAsync(base + o_ocean, in->ocean + sh, (size_t)P, hipMemcpyHostToDevice, s)); din.ocean = reinterpret_cast<const uint8_t*>(base + o_ocean); }
        if (out->diag) dout.diag = reinterpret_cast<uint16_t*>(base + o_diag);
        uint8_t** const d_u8[8] = {&dout.wtr1, &dout.wtr1_aerosol, &dout.wtr2, &dout.wtr, &dout.bwtr, &dout.conf, &dout.cloud,
                                   &dout.browse};
        for (int i = 0; i < 8; ++i) if (h_u8[i]) *d_u8[i] = reinterpret_cast<uint8_t*>(base + o_u8[i]);
        double** const d_f64[3] = {&dout.mndwi, &dout.ndvi, &dout.awesh};
        for (int i = 0; i < 3; ++i) if (h_f64[i]) *d_f64[i] = reinterpret_cast<double*>(base + o_f64[i]);
        int64_t* dcnt = counters ? reinterpret_cast<int64_t*>(base + o_cnt) : nullptr;
        int rc = dswx_classify_device_2d(ctx, params, 1, height, width, &din, &dout, dcnt, s);
        if (rc) return rc;
        if (out->diag) HIP_TRY(hipMemcpyAsync(out->diag + sh, dout.diag, (size_t)P * 2, hipMemcpyDeviceToHost, s));
        for (int i = 0; i < 8; ++i)
            if (h_u8[i]) HIP_TRY(hipMemcpyAsync(h_u8[i] + sh, *d_u8[i], (size_t)P, hipMemcpyDeviceToHost, s));
        for (int i = 0; i < 3; ++i)
            if (h_f64[i]) HIP_TRY(hipMemcpyAsync(h_f64[i] + sh, *d_f64[i], (size_t)P * 8, hipMemcpyDeviceToHost, s));
        if (counters) HIP_TRY(hipMemcpyAsync(counters + t * 3, dcnt, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return DSWX_OK;
}

}  // extern "C"
