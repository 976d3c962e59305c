// dswx_batch.hip -- resident batches: the planes of a batch allocated by the library (dswx_batch_create) and
// the opt-in measured placement of the output planes (dswx_batch_place_search).  include/dswx_hip.h
// "resident batches" documents the C-ABI; DESIGN.md section 5 has the measurements.
//
// Why placement at all.  On MI355X the rate of the fused kernel follows where its seven WRITE streams lie in
// the address space (reads stream at 7 TB/s anywhere; moving the INPUT planes by 170 GiB changes nothing).
// Moving the packed output region of a 256-tile batch (25.5 GiB) through one large allocation, the same launch
// runs at 0.70 - 0.74 of the HBM peak in some ranges and at 0.79 - 0.80 in others, with a structure of about
// 32 GiB (profiles/r03_write_stream_map.json: reproducible between fresh processes of one box for arenas
// >= 150 GB, similar but not equal between boxes).  Round 3 tried to turn that into a layout RULE (a fixed gap,
// a fixed offset, alternating output and input planes, separate allocations in either order) and none holds
// from one fresh process to the next for an allocation of the batch's own size: the same rule gives 0.70 - 0.80
// (profiles/r03_placement_rule_trials.json).  So the placement stays a measurement -- but one the library
// makes, behind the C-ABI, in two forms: dswx_batch_place_slide (the output planes in a range of the virtual
// address space that is longer than they are and backed chunk by chunk: ~100 candidate placements inside it,
// the chunks under the best one kept, the rest returned) and dswx_batch_place_search (one hipMalloc per output
// plane, each bound to the fastest of a few spare allocations).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "dswx_host.h"
#include "dswx_vmm.h"

using dswx_vmm::VaPool;
using dswx_vmm::VmRange;
using dswx_vmm::chunk_for;
using dswx_vmm::va_pool;

namespace {

constexpr uint64_t kAlign = 256;
inline uint64_t up256(uint64_t v) { return (v + (kAlign - 1)) & ~(kAlign - 1); }

struct PlaneSet {
    std::vector<int> in, out;    // plane indices present, in ABI order
};

PlaneSet planes_of(uint32_t flags) {
    PlaneSet s;
    for (int k = 0; k < 6; ++k) s.in.push_back(DSWX_PLANE_BAND0 + k);
    s.in.push_back(DSWX_PLANE_FMASK);
    if (flags & DSWX_BATCH_MASKS) {
        s.in.push_back(DSWX_PLANE_LAND);
        s.in.push_back(DSWX_PLANE_SHAD);
        s.in.push_back(DSWX_PLANE_OCEAN);
    }
    s.out.push_back(DSWX_PLANE_DIAG);
    s.out.push_back(DSWX_PLANE_WTR1);
    if (flags & DSWX_BATCH_WTR1_AEROSOL) s.out.push_back(DSWX_PLANE_WTR1_AEROSOL);
    s.out.push_back(DSWX_PLANE_WTR2);
    s.out.push_back(DSWX_PLANE_WTR);
    s.out.push_back(DSWX_PLANE_BWTR);
    s.out.push_back(DSWX_PLANE_CONF);
    s.out.push_back(DSWX_PLANE_CLOUD);
    if (flags & DSWX_BATCH_BROWSE) s.out.push_back(DSWX_PLANE_BROWSE);
    return s;
}

inline int elem_bytes(int plane) {
    return (plane <= DSWX_PLANE_BAND0 + 5 || plane == DSWX_PLANE_DIAG) ? 2 : 1;
}

int layout_mode(uint32_t flags, uint32_t* mode) {
    *mode = flags & (DSWX_BATCH_SEPARATE_OUTPUTS | DSWX_BATCH_SLIDING_OUTPUTS);
    if (*mode == (DSWX_BATCH_SEPARATE_OUTPUTS | DSWX_BATCH_SLIDING_OUTPUTS))
        return dswx_fail(DSWX_ERR_ARG, "DSWX_BATCH_SEPARATE_OUTPUTS and DSWX_BATCH_SLIDING_OUTPUTS exclude each other");
    return DSWX_OK;
}

}  // namespace

struct dswx_batch {
    dswx_ctx* ctx = nullptr;
    int device = -1;                           // of ctx; kept here so that destroy works after the context is gone
    dswx_batch_geom_t geom = {};
    uint32_t flags = 0;                        // as laid out (a sliding batch that fell back: without the sliding bit)
    uint32_t requested_flags = 0;              // as asked for
    std::string note;                          // why a sliding batch was allocated packed / a placement did nothing
    dswx_batch_layout_t lay = {};
    void* arena = nullptr;
    void* own[DSWX_BATCH_MAX_PLANES] = {};     // SEPARATE_OUTPUTS: the allocation an output plane lives in
    void* ptr[DSWX_BATCH_MAX_PLANES] = {};     // device address of every plane
    VmRange* range = nullptr;                  // SLIDING_OUTPUTS: the range the output region lives in ...
    size_t region_offset = 0;                  // ... and where in it
    int search_candidates = 0, search_probes = 0;
    float first_ms = 0.f, kept_ms = 0.f;
};

static void set_note(dswx_batch* b, const std::string& text) { b->note = text; }

static void bind_structs(const dswx_batch* b, dswx_planes_in_t* in, dswx_planes_out_t* out) {
    if (in) {
        memset(in, 0, sizeof *in);
        for (int k = 0; k < 6; ++k) in->band[k] = (const int16_t*)b->ptr[DSWX_PLANE_BAND0 + k];
        in->fmask = (const uint8_t*)b->ptr[DSWX_PLANE_FMASK];
        in->land = (const uint8_t*)b->ptr[DSWX_PLANE_LAND];
        in->shad = (const uint8_t*)b->ptr[DSWX_PLANE_SHAD];
        in->ocean = (const uint8_t*)b->ptr[DSWX_PLANE_OCEAN];
    }
    if (out) {
        memset(out, 0, sizeof *out);
        out->diag = (uint16_t*)b->ptr[DSWX_PLANE_DIAG];
        out->wtr1 = (uint8_t*)b->ptr[DSWX_PLANE_WTR1];
        out->wtr1_aerosol = (uint8_t*)b->ptr[DSWX_PLANE_WTR1_AEROSOL];
        out->wtr2 = (uint8_t*)b->ptr[DSWX_PLANE_WTR2];
        out->wtr = (uint8_t*)b->ptr[DSWX_PLANE_WTR];
        out->bwtr = (uint8_t*)b->ptr[DSWX_PLANE_BWTR];
        out->conf = (uint8_t*)b->ptr[DSWX_PLANE_CONF];
        out->cloud = (uint8_t*)b->ptr[DSWX_PLANE_CLOUD];
        out->browse = (uint8_t*)b->ptr[DSWX_PLANE_BROWSE];
    }
}

extern "C" {

int dswx_batch_layout(const dswx_batch_geom_t* geom, uint32_t flags, dswx_batch_layout_t* out) {
    if (!geom || !out) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (geom->n_tiles < 0 || geom->height < 0 || geom->width < 0 || geom->tile_stride < 0)
        return dswx_fail(DSWX_ERR_ARG, "negative size");
    uint32_t mode;
    if (int rc = layout_mode(flags, &mode)) return rc;
    if (flags & ~(uint32_t)(DSWX_BATCH_MASKS | DSWX_BATCH_WTR1_AEROSOL | DSWX_BATCH_BROWSE | DSWX_BATCH_SEPARATE_OUTPUTS |
                           DSWX_BATCH_SLIDING_OUTPUTS))
        return dswx_fail(DSWX_ERR_ARG, "unknown batch flag in 0x%x", flags);
    if (geom->height > (1LL << 30) || geom->width > (1LL << 30) || geom->n_tiles > (1LL << 32) ||
        geom->tile_stride > (1LL << 46))
        return dswx_fail(DSWX_ERR_ARG, "batch geometry out of range");
    memset(out, 0, sizeof *out);
    const int64_t P = geom->height * geom->width;
    int64_t stride = geom->tile_stride ? geom->tile_stride : (P + 255) / 256 * 256;
    if (stride < P) return dswx_fail(DSWX_ERR_ARG, "tile_stride smaller than the tile");
    out->tile_stride = stride;
    if (geom->n_tiles && (uint64_t)stride > (1ull << 46) / (uint64_t)geom->n_tiles)          // 64 TiB of pixels: no overflow below
        return dswx_fail(DSWX_ERR_ARG, "batch too large");
    const uint64_t px = (uint64_t)geom->n_tiles * (uint64_t)stride;
    const PlaneSet ps = planes_of(flags);
    for (int k : ps.in) out->plane_bytes[k] = up256(px * elem_bytes(k));
    for (int k : ps.out) out->plane_bytes[k] = up256(px * elem_bytes(k));
    out->plane_bytes[DSWX_PLANE_COUNTERS] = up256((uint64_t)geom->n_tiles * DSWX_N_COUNTERS * sizeof(int64_t));

    uint64_t cur = 0;
    auto take = [&](int k) {
        out->plane_offset[k] = cur;
        cur += out->plane_bytes[k] ? out->plane_bytes[k] : kAlign;      // an empty batch still has distinct addresses
    };
    for (int k : ps.in) take(k);
    if (mode == 0)
        for (int k : ps.out) take(k);
    take(DSWX_PLANE_COUNTERS);
    out->arena_bytes = cur;
    if (mode == DSWX_BATCH_SLIDING_OUTPUTS) {       // offsets inside the output REGION, which lives in a range of its own
        cur = 0;
        for (int k : ps.out) take(k);
        out->write_span_bytes = cur;
    }
    if (mode == 0) {
        uint64_t lo = ~0ull, hi = 0;
        for (int k : ps.out) {
            lo = out->plane_offset[k] < lo ? out->plane_offset[k] : lo;
            const uint64_t end = out->plane_offset[k] + out->plane_bytes[k];
            hi = end > hi ? end : hi;
        }
        out->write_span_bytes = hi - lo;
    }
    return DSWX_OK;
}

int dswx_batch_destroy(dswx_batch_t* b) {
    if (!b) return DSWX_OK;
    if (b->device >= 0) (void)hipSetDevice(b->device);
    for (void* p : b->own)
        if (p) (void)hipFree(p);
    if (b->arena) (void)hipFree(b->arena);
    if (b->range) { b->range->destroy(); delete b->range; }
    delete b;
    return DSWX_OK;
}

int dswx_batch_create(dswx_ctx_t* ctx, const dswx_batch_geom_t* geom, uint32_t flags, dswx_batch_t** out) {
    if (!ctx || !geom || !out) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    *out = nullptr;
    dswx_batch_layout_t lay;
    if (int rc = dswx_batch_layout(geom, flags, &lay)) return rc;
    uint32_t mode;
    if (int rc = layout_mode(flags, &mode)) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if (mode == DSWX_BATCH_SLIDING_OUTPUTS) {
        int vmm = 0;
        if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, ctx->device) != hipSuccess || !vmm)
            return dswx_fail(DSWX_ERR_UNSUPPORTED, "DSWX_BATCH_SLIDING_OUTPUTS needs HIP virtual memory management, which device "
                             "%d does not report; use DSWX_BATCH_SEPARATE_OUTPUTS + dswx_batch_place_search", ctx->device);
    }
    dswx_batch* b = new dswx_batch();
    b->ctx = ctx;
    b->device = ctx->device;
    b->geom = *geom;
    b->requested_flags = flags;
    hipError_t e = hipSuccess;
    if (mode == DSWX_BATCH_SLIDING_OUTPUTS) {
        // the range first: if the address space cannot be had (the reservation refused, or the library's budget of
        // retired addresses spent -- dswx_batch_va_budget), the batch is allocated PACKED instead, with the reason on
        // record: a long-lived service degrades to the first-come rate, it does not fail
        const size_t bytes = lay.write_span_bytes ? lay.write_span_bytes : kAlign;
        b->range = new VmRange();
        if (b->range->create(ctx->device, bytes, chunk_for(bytes)) != hipSuccess) {
            (void)hipGetLastError();
            set_note(b, "DSWX_BATCH_SLIDING_OUTPUTS not honoured, planes packed in one allocation: " + b->range->why);
            delete b->range;
            b->range = nullptr;
            flags &= ~(uint32_t)DSWX_BATCH_SLIDING_OUTPUTS;
            mode = 0;
            if (int rc = dswx_batch_layout(geom, flags, &lay)) { delete b; return rc; }
        }
    }
    b->geom.tile_stride = lay.tile_stride;
    b->flags = flags;
    b->lay = lay;
    e = dswx_locked_malloc(&b->arena, lay.arena_bytes);
    const PlaneSet ps = planes_of(flags);
    if (e == hipSuccess && mode == DSWX_BATCH_SEPARATE_OUTPUTS)
        for (int k : ps.out) {
            e = dswx_locked_malloc(&b->own[k], lay.plane_bytes[k] ? lay.plane_bytes[k] : kAlign);
            if (e != hipSuccess) break;
        }
    if (e != hipSuccess) {
        dswx_batch_destroy(b);
        (void)hipGetLastError();
        uint64_t pooled = 0;
        dswx_vmm::account(0, nullptr, nullptr, nullptr, nullptr, &pooled);
        // (the library does not trim by itself: dswx_batch_pool_trim has a window in which other threads must not allocate)
        return dswx_fail(DSWX_ERR_HIP, "dswx_batch_create: device allocation failed: %s (arena of %llu bytes; the library's pool of "
                         "placement chunks holds %llu bytes that dswx_batch_pool_trim() returns to the device)",
                         hipGetErrorString(e), (unsigned long long)lay.arena_bytes, (unsigned long long)pooled);
    }
    for (int k : ps.in) b->ptr[k] = (char*)b->arena + lay.plane_offset[k];
    for (int k : ps.out)
        b->ptr[k] = b->own[k] ? b->own[k] : (b->range ? b->range->va : (char*)b->arena) + lay.plane_offset[k];
    b->ptr[DSWX_PLANE_COUNTERS] = (char*)b->arena + lay.plane_offset[DSWX_PLANE_COUNTERS];
    *out = b;
    return DSWX_OK;
}

int dswx_batch_va_budget(uint64_t new_budget_bytes, uint64_t* budget_bytes, uint64_t* live_bytes, uint64_t* retired_bytes,
                         uint64_t* loose_bytes, uint64_t* pooled_bytes) {
    dswx_vmm::account(new_budget_bytes, budget_bytes, live_bytes, retired_bytes, loose_bytes, pooled_bytes);
    return DSWX_OK;
}

int dswx_batch_pool_trim(uint64_t* released_bytes) {
    const uint64_t released = dswx_vmm::pool_trim();
    if (released_bytes) *released_bytes = released;
    return DSWX_OK;
}

int dswx_batch_planes(const dswx_batch_t* b, dswx_batch_geom_t* geom, dswx_planes_in_t* in, dswx_planes_out_t* out,
                      int64_t** counters) {
    if (!b) return dswx_fail(DSWX_ERR_ARG, "batch is NULL");
    if (geom) *geom = b->geom;
    bind_structs(b, in, out);
    if (counters) *counters = (int64_t*)b->ptr[DSWX_PLANE_COUNTERS];
    return DSWX_OK;
}

int dswx_batch_info(const dswx_batch_t* b, dswx_batch_info_t* info) {
    if (!b || !info) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    memset(info, 0, sizeof *info);
    info->geom = b->geom;
    info->flags = b->flags;
    info->n_allocations = 1;
    info->bytes_allocated = b->lay.arena_bytes;
    for (int k = 0; k < DSWX_BATCH_MAX_PLANES; ++k)
        if (b->own[k]) {
            ++info->n_allocations;
            info->bytes_allocated += b->lay.plane_bytes[k] ? b->lay.plane_bytes[k] : kAlign;
        }
    if (b->range) {
        ++info->n_allocations;
        info->bytes_allocated += b->range->mapped_bytes();
        info->va_reserved_bytes = b->range->reserved;
    }
    {
        VaPool& pool = va_pool();
        std::lock_guard<std::mutex> lock(dswx_va_mutex());
        info->va_retired_bytes = pool.retired;
        info->va_budget_bytes = pool.budget;
        info->va_pooled_bytes = pool.pooled;
    }
    snprintf(info->note, sizeof info->note, "%s", b->note.c_str());
    info->search_candidates = b->search_candidates;
    info->search_probes = b->search_probes;
    info->first_come_launch_ms = b->first_ms;
    info->kept_launch_ms = b->kept_ms;
    return DSWX_OK;
}

int dswx_batch_classify(dswx_batch_t* b, const dswx_params_t* params, int64_t n_tiles, void* stream) {
    if (!b) return dswx_fail(DSWX_ERR_ARG, "batch is NULL");
    if (n_tiles == DSWX_BATCH_ALL_TILES) n_tiles = b->geom.n_tiles;
    if (n_tiles < 0 || n_tiles > b->geom.n_tiles)
        return dswx_fail(DSWX_ERR_ARG, "n_tiles %lld outside the batch (%lld resident)", (long long)n_tiles,
                         (long long)b->geom.n_tiles);
    if (n_tiles == 0) return DSWX_OK;          // an empty chunk is no work (not "all": a walk's empty last chunk must not re-classify the batch)
    dswx_batch_geom_t g = b->geom;
    g.n_tiles = n_tiles;
    dswx_planes_in_t in;
    dswx_planes_out_t out;
    bind_structs(b, &in, &out);
    return dswx_classify_batch(b->ctx, params, &g, &in, &out, (int64_t*)b->ptr[DSWX_PLANE_COUNTERS], stream);
}

int dswx_batch_synth(dswx_batch_t* b, uint64_t seed, int64_t tile0, void* stream) {
    if (!b) return dswx_fail(DSWX_ERR_ARG, "batch is NULL");
    dswx_planes_in_t in;
    bind_structs(b, &in, nullptr);
    return dswx_synth_batch(b->ctx, seed, tile0, &b->geom, &in, stream);
}

// `launches` launches of the real kernel over the whole batch, after one untimed launch; ms per launch
static int probe_ms(dswx_batch* b, const dswx_params_t* params, int launches, hipEvent_t e0, hipEvent_t e1, float* ms) {
    hipStream_t s = b->ctx->stream;
    if (int rc = dswx_batch_classify(b, params, DSWX_BATCH_ALL_TILES, nullptr)) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < launches; ++i)
        if (int rc = dswx_batch_classify(b, params, DSWX_BATCH_ALL_TILES, nullptr)) return rc;
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(ms, e0, e1));
    *ms /= (float)launches;
    return DSWX_OK;
}

int dswx_batch_place_search(dswx_batch_t* b, const dswx_params_t* params, int32_t candidates, int32_t launches,
                            uint64_t keep_free_bytes) {
    if (!b || !params) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (!(b->flags & DSWX_BATCH_SEPARATE_OUTPUTS))
        return dswx_fail(DSWX_ERR_ARG, "dswx_batch_place_search needs a DSWX_BATCH_SEPARATE_OUTPUTS batch");
    if (candidates < 1 || launches < 1) return dswx_fail(DSWX_ERR_ARG, "candidates and launches must be positive");
    HIP_TRY(hipSetDevice(b->ctx->device));
    const PlaneSet ps = planes_of(b->flags);
    uint64_t out_bytes = 0;
    for (int k : ps.out) out_bytes += b->lay.plane_bytes[k] ? b->lay.plane_bytes[k] : kAlign;

    // spare sets, side by side (a freed range would simply be handed out again), bounded by free memory
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    int sets = candidates - 1;
    const uint64_t room = free_b > keep_free_bytes ? free_b - keep_free_bytes : 0;
    if ((uint64_t)sets * out_bytes > room) sets = (int)(room / (out_bytes ? out_bytes : 1));
    std::vector<std::vector<void*>> spare(DSWX_BATCH_MAX_PLANES);      // per plane: candidates of ITS size
    bool full = true;
    int complete_sets = 0;
    for (int sidx = 0; sidx < sets && full; ++sidx) {
        for (int k : ps.out) {
            void* p = nullptr;
            if (dswx_locked_malloc(&p, b->lay.plane_bytes[k] ? b->lay.plane_bytes[k] : kAlign) != hipSuccess) {
                (void)hipGetLastError();        // refused: search among what there is
                full = false;
                break;
            }
            spare[k].push_back(p);
        }
        complete_sets += full ? 1 : 0;
    }
    // planes of equal size share a pool: a candidate one plane did not take is a candidate for the next
    auto pool_of = [&](int k) -> std::vector<void*>& {
        for (int j : ps.out)
            if (b->lay.plane_bytes[j] == b->lay.plane_bytes[k]) return spare[j];
        return spare[k];
    };
    for (int k : ps.out) {
        std::vector<void*>& pool = pool_of(k);
        if (&pool != &spare[k]) {
            pool.insert(pool.end(), spare[k].begin(), spare[k].end());
            spare[k].clear();
        }
    }
    void* original[DSWX_BATCH_MAX_PLANES];
    memcpy(original, b->own, sizeof original);

    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = DSWX_OK;
    int probes = 0;
    float first_ms = 0.f, kept_ms = 0.f;
    auto bind = [&](int k, void* p) { b->own[k] = p; b->ptr[k] = p; };
    do {
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            rc = dswx_fail(DSWX_ERR_HIP, "hipEventCreate failed");
            break;
        }
        if ((rc = probe_ms(b, params, launches, e0, e1, &first_ms))) break;
        kept_ms = first_ms;
        for (int k : ps.out) {
            std::vector<void*>& pool = pool_of(k);
            if (pool.empty()) continue;
            float best_ms;
            if ((rc = probe_ms(b, params, launches, e0, e1, &best_ms))) break;
            int best = -1;
            void* mine = b->own[k];
            for (size_t c = 0; c < pool.size(); ++c) {
                bind(k, pool[c]);
                float ms;
                rc = probe_ms(b, params, launches, e0, e1, &ms);
                bind(k, mine);
                if (rc) break;
                ++probes;
                if (ms < best_ms) { best_ms = ms; best = (int)c; }
            }
            if (rc) break;
            if (best >= 0) {
                bind(k, pool[best]);
                pool[best] = mine;
            }
        }
        if (rc || !probes) break;
        // judge the outcome under equal conditions (the part is warmer now than at first_ms): the chosen planes
        // and the first-come planes back to back; keep the better set
        void* chosen[DSWX_BATCH_MAX_PLANES];
        memcpy(chosen, b->own, sizeof chosen);
        float chosen_ms, again_ms;
        if ((rc = probe_ms(b, params, launches, e0, e1, &chosen_ms))) break;
        for (int k : ps.out) bind(k, original[k]);
        if ((rc = probe_ms(b, params, launches, e0, e1, &again_ms))) break;
        first_ms = again_ms;
        if (chosen_ms < again_ms) {
            for (int k : ps.out) bind(k, chosen[k]);
            kept_ms = chosen_ms;
        } else {
            kept_ms = again_ms;
        }
    } while (false);
    if (rc)          // whatever happened, the planes a caller may already hold stay valid
        for (int k : ps.out) bind(k, original[k]);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    // exactly the allocations the planes point at stay alive
    (void)hipStreamSynchronize(b->ctx->stream);
    std::vector<void*> all;
    for (int k : ps.out) {
        all.push_back(original[k]);
        all.insert(all.end(), spare[k].begin(), spare[k].end());
    }
    std::sort(all.begin(), all.end());       // a first-come plane that lost its place sits in a pool as well
    all.erase(std::unique(all.begin(), all.end()), all.end());
    for (void* p : all) {
        bool bound = false;
        for (int k : ps.out) bound = bound || b->own[k] == p;
        if (!bound) (void)hipFree(p);
    }
    if (rc) return rc;
    b->search_candidates = complete_sets + 1;      // complete spare sets actually obtained + the first-come planes
    b->search_probes = probes;
    b->first_ms = first_ms;
    b->kept_ms = kept_ms;
    return DSWX_OK;
}

// Sliding placement: see include/dswx_hip.h.
int dswx_batch_place_slide(dswx_batch_t* b, const dswx_params_t* params, uint64_t slack_bytes, uint64_t step_bytes,
                           int32_t spread_gaps, int32_t refine_passes, int32_t launches, uint64_t keep_free_bytes) {
    if (!b || !params) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (!(b->requested_flags & DSWX_BATCH_SLIDING_OUTPUTS))
        return dswx_fail(DSWX_ERR_ARG, "dswx_batch_place_slide needs a DSWX_BATCH_SLIDING_OUTPUTS batch");
    if (launches < 1 || step_bytes == 0 || spread_gaps < 0 || refine_passes < 0)
        return dswx_fail(DSWX_ERR_ARG, "launches and step_bytes must be positive, spread_gaps / refine_passes not negative");
    HIP_TRY(hipSetDevice(b->ctx->device));
    if (!b->range) {
        // the batch was asked for as a sliding one and allocated packed (dswx_batch_create's fallback): there is nothing
        // to slide; time the planes as they are so that the record is complete, keep the note
        hipEvent_t t0 = nullptr, t1 = nullptr;
        int prc = DSWX_OK;
        float ms = 0.f;
        if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) prc = dswx_fail(DSWX_ERR_HIP, "hipEventCreate failed");
        else prc = probe_ms(b, params, launches, t0, t1, &ms);
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
        if (prc) return prc;
        b->search_candidates = b->search_probes = 0;
        b->first_ms = b->kept_ms = ms;
        return DSWX_OK;
    }
    const PlaneSet ps = planes_of(b->flags);
    const size_t region = b->lay.write_span_bytes ? b->lay.write_span_bytes : kAlign;
    const size_t step = (size_t)((step_bytes + 255) & ~(uint64_t)255);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    // the wider range is held BESIDE the current one while the search runs
    uint64_t usable = free_b;          // + the pooled chunks of the size the wide range is built from
    {
        std::lock_guard<std::mutex> lock(dswx_va_mutex());
        for (const VaPool::Spare& sp : va_pool().spare)
            if (sp.device == b->ctx->device && sp.chunk == chunk_for(region)) usable += sp.chunk;
    }
    const uint64_t room = usable > keep_free_bytes + region ? usable - keep_free_bytes - region : 0;
    size_t slack = (size_t)(slack_bytes < room ? slack_bytes : room);
    slack = slack / step * step;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    VmRange* wide = nullptr;
    void* first_ptr[DSWX_BATCH_MAX_PLANES];
    memcpy(first_ptr, b->ptr, sizeof first_ptr);
    // a candidate: the output planes in their order, the first at `off`, `gap` bytes of distance added between
    // consecutive planes (0 = packed)
    auto bind_at = [&](char* base, size_t off, size_t gap) {
        size_t n = 0;
        for (int k : ps.out) b->ptr[k] = base + off + b->lay.plane_offset[k] + gap * n++;
    };
    int rc = DSWX_OK, positions = 0;
    float first_ms = 0.f, kept_ms = 0.f;
    do {
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            rc = dswx_fail(DSWX_ERR_HIP, "hipEventCreate failed");
            break;
        }
        if ((rc = probe_ms(b, params, launches, e0, e1, &first_ms))) break;
        kept_ms = first_ms;
        if (slack == 0) {                           // no room to slide in: the planes stay where they are
            set_note(b, "dswx_batch_place_slide: no device memory to slide in, planes left where they are");
            break;
        }
        wide = new VmRange();
        const size_t chunk = chunk_for(region);
        hipError_t e = wide->create(b->ctx->device, region + slack, chunk);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_note(b, "dswx_batch_place_slide: planes left where they are: " + wide->why);
            delete wide;
            wide = nullptr;
            break;                                  // refused: nothing to choose from
        }
        size_t best_off = 0, best_gap = 0;
        float best_ms = 0.f;
        // lab switch (tests): the candidate with this index is the one kept, whatever the clock says
        const int force = b->ctx->place_force_candidate;
        auto consider = [&](size_t off, size_t gap) {
            bind_at(wide->va, off, gap);
            float ms;
            if ((rc = probe_ms(b, params, launches, e0, e1, &ms))) return;
            if (positions == 0 || (force >= 0 ? positions == force : ms < best_ms)) { best_ms = ms; best_off = off; best_gap = gap; }
            ++positions;
        };
        // (1) the packed region at every step of the range
        for (size_t off = 0; off + region <= wide->reserved && !rc; off += step) consider(off, 0);
        // (2) the planes spread over the range with equal gaps (each write stream in a neighbourhood of its own),
        //     if the range is long enough for gaps of at least one step
        const size_t n_out = ps.out.size();
        if (!rc && spread_gaps > 0 && n_out > 1) {
            const size_t max_gap = (wide->reserved - region) / (n_out - 1) / step * step;
            for (int g = 1; g <= spread_gaps && !rc; ++g) {
                const size_t gap = max_gap * (size_t)g / (size_t)spread_gaps / step * step;
                if (gap) consider(0, gap);
            }
        }
        if (rc) break;
        // (3) refinement: from the best candidate, every plane in turn (DIAG first) tries the other free places of the
        //     range on a coarser grid and keeps the one under which the launch runs fastest -- the per-plane freedom of
        //     dswx_batch_place_search without its spare allocations (the range is already mapped)
        std::vector<size_t> pos(n_out), len(n_out);
        {
            size_t n = 0;
            for (int k : ps.out) {
                pos[n] = best_off + b->lay.plane_offset[k] + best_gap * n;
                len[n] = b->lay.plane_bytes[k] ? b->lay.plane_bytes[k] : kAlign;
                ++n;
            }
        }
        char* const wide_va = wide->va;          // (`wide` itself is handed to the batch below)
        auto bind_pos = [&]() {
            size_t n = 0;
            for (int k : ps.out) b->ptr[k] = wide_va + pos[n++];
        };
        for (int pass = 0; pass < refine_passes && force < 0 && !rc; ++pass) {
            const size_t rstep = step * 2;
            for (size_t i = 0; i < n_out && !rc; ++i) {
                bind_pos();
                float here_ms;
                if ((rc = probe_ms(b, params, launches, e0, e1, &here_ms))) break;
                size_t best_q = pos[i];
                const size_t mine = pos[i];
                for (size_t q = 0; q + len[i] <= wide->reserved && !rc; q += rstep) {
                    bool clash = false;
                    for (size_t j = 0; j < n_out; ++j)
                        clash = clash || (j != i && q < pos[j] + len[j] && pos[j] < q + len[i]);
                    if (clash || q == mine) continue;
                    pos[i] = q;
                    bind_pos();
                    float ms;
                    if ((rc = probe_ms(b, params, launches, e0, e1, &ms))) break;
                    ++positions;
                    if (ms < here_ms) { here_ms = ms; best_q = q; }
                }
                pos[i] = best_q;
            }
        }
        if (rc) break;
        // equal conditions (the part is warmer now): the best candidate and the first-come range back to back
        float chosen_ms, again_ms;
        bind_pos();
        if ((rc = probe_ms(b, params, launches, e0, e1, &chosen_ms))) break;
        memcpy(b->ptr, first_ptr, sizeof first_ptr);
        if ((rc = probe_ms(b, params, launches, e0, e1, &again_ms))) break;
        first_ms = again_ms;
        if (force >= 0 || chosen_ms < again_ms) {
            // keep the chosen placement: its chunks -- the physical memory -- move into a fresh range (addresses no kernel
            // has used), the wide range is dropped whole, which is what gives its other chunks back to the device
            (void)hipStreamSynchronize(b->ctx->stream);
            std::vector<std::pair<size_t, size_t>> keep;
            for (size_t i = 0; i < n_out; ++i) keep.push_back({pos[i], pos[i] + len[i]});
            size_t base = 0;
            VmRange* home = wide->rehome(keep, &base);
            if (!home) {
                set_note(b, "dswx_batch_place_slide: the chosen chunks could not be moved (" + wide->why + "), planes left where they were");
                kept_ms = again_ms;
            } else {
                size_t n = 0;
                for (int k : ps.out) b->ptr[k] = home->va + (pos[n++] - base);
                float homed_ms;
                if ((rc = probe_ms(b, params, launches, e0, e1, &homed_ms))) {
                    memcpy(b->ptr, first_ptr, sizeof first_ptr);
                    (void)hipStreamSynchronize(b->ctx->stream);
                    home->destroy();
                    delete home;
                    break;
                }
                if (force >= 0 || homed_ms < again_ms) {
                    (void)hipStreamSynchronize(b->ctx->stream);
                    b->range->destroy();
                    delete b->range;
                    b->range = home;
                    b->region_offset = pos[0] - base;
                    kept_ms = homed_ms;
                } else {            // under its new addresses the placement is no better than the first-come one: drop it
                    memcpy(b->ptr, first_ptr, sizeof first_ptr);
                    (void)hipStreamSynchronize(b->ctx->stream);
                    home->destroy();
                    delete home;
                    kept_ms = again_ms;
                }
            }
        } else {
            kept_ms = again_ms;
        }
    } while (false);
    if (rc) memcpy(b->ptr, first_ptr, sizeof first_ptr);
    (void)hipStreamSynchronize(b->ctx->stream);
    if (wide) { wide->destroy(); delete wide; }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc) return rc;
    b->search_candidates = positions;
    b->search_probes = positions;
    b->first_ms = first_ms;
    b->kept_ms = kept_ms;
    return DSWX_OK;
}

}  // extern "C"
