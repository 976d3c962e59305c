/*
 * resident_batch.c -- the C-ABI of include/dswx_hip.h from plain C (no Python, no HIP headers):
 * a resident batch of synthetic HLS tiles is allocated by the library, its output planes are placed by
 * measurement, it is classified, and the per-tile coverage counters (dswx_hls.py:5104-5136) and a
 * checksum of every layer are printed.
 *
 *   gcc -std=c11 -O2 -I include examples/resident_batch.c -L proteus_amd/_lib -ldswx_hip \
 *       -Wl,-rpath,$PWD/proteus_amd/_lib -o resident_batch && ./resident_batch [n_tiles] [size]
 *
 * tests/test_integration_stub.py::test_c_example_builds_and_runs compiles it with gcc, runs it on the GPU and
 * compares counters and checksums with the oracle's.
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dswx_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc__ = (call);                                                                 \
        if (rc__ != DSWX_OK) {                                                             \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, dswx_last_error());       \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

/* FNV-1a over the first `pixels` elements of every tile of a downloaded plane */
static uint64_t checksum(const uint8_t* host, int64_t n_tiles, int64_t stride_bytes, int64_t tile_bytes) {
    uint64_t h = 1469598103934665603ull;
    for (int64_t t = 0; t < n_tiles; ++t)
        for (int64_t i = 0; i < tile_bytes; ++i) {
            h ^= host[t * stride_bytes + i];
            h *= 1099511628211ull;
        }
    return h;
}

int main(int argc, char** argv) {
    const int64_t n_tiles = argc > 1 ? atoll(argv[1]) : 4;
    const int64_t size = argc > 2 ? atoll(argv[2]) : 512;
    if (dswx_abi_version() != DSWX_ABI_VERSION) {
        fprintf(stderr, "header / library ABI mismatch: %d vs %d\n", DSWX_ABI_VERSION, dswx_abi_version());
        return 1;
    }
    dswx_ctx_t* ctx = NULL;
    CHECK(dswx_ctx_create(0, &ctx));               /* DSWX_ERR_NO_DEVICE without an MI355X: there is no CPU fallback */
    dswx_params_t params;
    CHECK(dswx_params_default(&params));            /* defaults/dswx_hls.yaml */

    dswx_batch_geom_t geom = {n_tiles, size, size, 0};          /* stride 0: padded to 256 pixels by the library */
    dswx_batch_t* batch = NULL;
    CHECK(dswx_batch_create(ctx, &geom, DSWX_BATCH_MASKS | DSWX_BATCH_SLIDING_OUTPUTS, &batch));
    CHECK(dswx_batch_synth(batch, 20251010u, 0, NULL));         /* a real caller uploads its tiles into the planes */
    /* place the output planes: 16 MiB of slack in 2 MiB steps is plenty for a toy batch (bench.py: 48 GiB / 2 GiB) */
    CHECK(dswx_batch_place_slide(batch, &params, 16u << 20, 2u << 20, 2, 1, 2, 0));
    CHECK(dswx_batch_classify(batch, &params, DSWX_BATCH_ALL_TILES, NULL));
    CHECK(dswx_stream_synchronize(ctx, NULL));

    dswx_planes_out_t out;
    int64_t* d_counters = NULL;
    CHECK(dswx_batch_planes(batch, &geom, NULL, &out, &d_counters));
    dswx_batch_info_t info;
    CHECK(dswx_batch_info(batch, &info));
    printf("tiles %" PRId64 " of %" PRId64 " x %" PRId64 ", tile stride %" PRId64 " px, %d allocations, %" PRIu64
           " bytes, %d placements probed\n", geom.n_tiles, geom.height, geom.width, geom.tile_stride, info.n_allocations,
           info.bytes_allocated, info.search_probes);
    /* ABI v5: what the placement cost in ADDRESS SPACE (never memory), and why it did nothing if it did nothing */
    uint64_t budget = 0, live = 0, retired = 0, loose = 0, pooled = 0;
    CHECK(dswx_batch_va_budget(0, &budget, &live, &retired, &loose, &pooled));
    printf("address space: %" PRIu64 " bytes reserved by this batch, %" PRIu64 " retired, %" PRIu64 " loose, budget %" PRIu64
           "; %" PRIu64 " bytes of memory pooled%s%s\n", info.va_reserved_bytes, retired, loose, budget, pooled,
           info.note[0] ? "; note: " : "", info.note);

    int64_t* counters = malloc((size_t)n_tiles * DSWX_N_COUNTERS * sizeof *counters);
    if (!counters) return 1;
    CHECK(dswx_memcpy_d2h(ctx, counters, d_counters, (size_t)n_tiles * DSWX_N_COUNTERS * sizeof *counters));
    for (int64_t t = 0; t < n_tiles; ++t)
        printf("counters %" PRId64 ": n_valid %" PRId64 " n_cloud_and_valid %" PRId64 " n_not_ocean %" PRId64 "\n", t,
               counters[t * 3 + DSWX_N_VALID], counters[t * 3 + DSWX_N_CLOUD_AND_VALID], counters[t * 3 + DSWX_N_NOT_OCEAN]);

    const struct { const char* name; const void* ptr; int elem; } layers[] = {
        {"diag", out.diag, 2}, {"wtr1", out.wtr1, 1}, {"wtr2", out.wtr2, 1}, {"wtr", out.wtr, 1},
        {"bwtr", out.bwtr, 1}, {"conf", out.conf, 1}, {"cloud", out.cloud, 1}};
    const size_t plane_bytes = (size_t)n_tiles * (size_t)geom.tile_stride * 2;
    uint8_t* host = malloc(plane_bytes);
    if (!host) return 1;
    for (size_t k = 0; k < sizeof layers / sizeof layers[0]; ++k) {
        const size_t nbytes = (size_t)n_tiles * (size_t)geom.tile_stride * (size_t)layers[k].elem;
        CHECK(dswx_memcpy_d2h(ctx, host, layers[k].ptr, nbytes));
        printf("checksum %s %016" PRIx64 "\n", layers[k].name,
               checksum(host, n_tiles, geom.tile_stride * layers[k].elem, size * size * layers[k].elem));
    }
    free(host);
    free(counters);
    CHECK(dswx_batch_destroy(batch));
    CHECK(dswx_batch_pool_trim(NULL));             /* single-threaded here: the pooled chunks back to the device */
    CHECK(dswx_ctx_destroy(ctx));
    return 0;
}
