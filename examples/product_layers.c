/*
 * product_layers.c -- the writer / reader side of the C-ABI (ABI v6) from plain C: one synthetic tile is classified in a
 * resident batch, its WTR layer leaves as the 2-D blocks of a cloud-optimized GeoTIFF -- full resolution + NEAREST
 * overviews 4 / 16 / 64 / 128, PREDICTOR=2 applied, what save_as_cog (src/proteus/core.py:7-91) hands to DEFLATE -- made
 * on the device (dswx_cog_layout, dswx_cog_blocks_device), and the full-resolution blocks go back through the reader's
 * kernel (dswx_untile_device) into the plane they came from.
 *
 *   gcc -std=c11 -O2 -I include examples/product_layers.c -L proteus_amd/_lib -ldswx_hip \
 *       -Wl,-rpath,$PWD/proteus_amd/_lib -o product_layers && ./product_layers [size] [tile]
 *
 * tests/test_integration_stub.py compiles it with gcc -Wall -Wextra -Werror, runs it on the GPU and compares the level
 * geometry and the checksum of every level's bytes with oracle/cog_oracle.py's row-by-row restatement.
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

static uint64_t fnv1a(const uint8_t* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}

int main(int argc, char** argv) {
    const int64_t size = argc > 1 ? atoll(argv[1]) : 700;
    const int32_t tile = argc > 2 ? atoi(argv[2]) : 128;
    const int32_t factors[4] = {4, 16, 64, 128};
    dswx_ctx_t* ctx = NULL;
    CHECK(dswx_ctx_create(0, &ctx));               /* DSWX_ERR_NO_DEVICE without an MI355X: there is no CPU fallback */
    dswx_params_t params;
    CHECK(dswx_params_default(&params));

    /* one tile, contiguous (tile stride = size * size), classified where it lies */
    dswx_batch_geom_t geom = {1, size, size, size * size};
    dswx_batch_t* batch = NULL;
    CHECK(dswx_batch_create(ctx, &geom, DSWX_BATCH_MASKS, &batch));
    CHECK(dswx_batch_synth(batch, 20251010u, 0, NULL));
    CHECK(dswx_batch_classify(batch, &params, DSWX_BATCH_ALL_TILES, NULL));
    dswx_planes_out_t out;
    CHECK(dswx_batch_planes(batch, &geom, NULL, &out, NULL));

    /* WTR -> blocks of every level */
    dswx_cog_layout_t lay;
    CHECK(dswx_cog_layout(size, size, 1, tile, factors, 4, &lay));
    void* d_blocks = NULL;
    CHECK(dswx_device_malloc(ctx, (size_t)lay.total_bytes, &d_blocks));
    CHECK(dswx_cog_blocks_device(ctx, out.wtr, 1, size, size, tile, factors, 4, 2, d_blocks, NULL));
    CHECK(dswx_stream_synchronize(ctx, NULL));
    uint8_t* blocks = malloc((size_t)lay.total_bytes);
    if (!blocks) return 1;
    CHECK(dswx_memcpy_d2h(ctx, blocks, d_blocks, (size_t)lay.total_bytes));
    printf("levels %d, %" PRIu64 " bytes\n", lay.n_levels, lay.total_bytes);
    for (int k = 0; k < lay.n_levels; ++k) {
        const size_t n = (size_t)lay.blocks_down[k] * (size_t)lay.blocks_across[k] * (size_t)tile * (size_t)tile;
        printf("level %d: factor %d, %" PRId64 " x %" PRId64 ", %d x %d blocks, checksum %016" PRIx64 "\n", k, lay.factor[k],
               lay.height[k], lay.width[k], lay.blocks_down[k], lay.blocks_across[k], fnv1a(blocks + lay.offset_bytes[k], n));
    }

    /* the reader's direction: the full-resolution blocks -> a plane; it must be the layer */
    void* d_plane = NULL;
    CHECK(dswx_device_malloc(ctx, (size_t)(size * size), &d_plane));
    CHECK(dswx_untile_device(ctx, d_blocks, 1, size, size, tile, tile, 2, d_plane, NULL));
    CHECK(dswx_stream_synchronize(ctx, NULL));
    uint8_t* layer = malloc((size_t)(size * size));
    uint8_t* back = malloc((size_t)(size * size));
    if (!layer || !back) return 1;
    CHECK(dswx_memcpy_d2h(ctx, layer, out.wtr, (size_t)(size * size)));
    CHECK(dswx_memcpy_d2h(ctx, back, d_plane, (size_t)(size * size)));
    printf("layer checksum %016" PRIx64 ", round trip %s\n", fnv1a(layer, (size_t)(size * size)),
           memcmp(layer, back, (size_t)(size * size)) == 0 ? "ok" : "DIFFERS");

    free(back);
    free(layer);
    free(blocks);
    CHECK(dswx_device_free(ctx, d_plane));
    CHECK(dswx_device_free(ctx, d_blocks));
    CHECK(dswx_batch_destroy(batch));
    CHECK(dswx_ctx_destroy(ctx));
    return 0;
}
