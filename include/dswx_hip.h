/*
 * dswx_hip.h -- C ABI of the MI355X (gfx950) DSWx-HLS per-pixel classifier.
 *
 * This is the drop-in boundary for the per-pixel hot path of NASA PROTEUS
 * `src/proteus/dswx_hls.py` (reference file:line given per entry point).  The
 * reference has no FFI: its seam is the run of numpy calls inside
 * generate_dswx_layers (:5088-5112, :5225-5286, :5358-5369).  A maintainer binds
 * this library from Python with ctypes (INTEGRATION.md shows the stub) and
 * replaces that run of calls with ONE call to dswx_classify_host().
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer; no
 * allocation crosses the ABI except through dswx_device_malloc/free; every
 * function returns DSWX_OK (0) or a negative status and records a message
 * retrievable with dswx_last_error() (thread-local).  There is no CPU fallback:
 * without a HIP device dswx_ctx_create() fails with DSWX_ERR_NO_DEVICE.
 *
 * Plane layout ("band-planar batch"): every plane is [n_tiles][tile_stride] with
 * the tile's H*W pixels row-major at the start of its slot.  tile_stride defaults
 * to H*W (contiguous tiles, what the reference's seam holds).  For the full HBM rate
 * make every plane BASE 256-byte aligned (a hipMalloc pointer is) and the tile stride
 * a multiple of 8 pixels: the table-driven kernel then keeps every wave access on a
 * 128-byte line boundary on any such stride -- a 3660 x 3660 tile is 13,395,600
 * pixels = 144 mod 256, and a fixed thread -> pixel mapping would put every 1 KiB wave
 * store of tiles 1.. across partial lines (5.1 - 5.6 vs 5.7 - 6.3 TB/s, DESIGN.md
 * section 5); since ABI v5 the kernel shifts the mapping per tile instead, and a
 * stride padded to 256 pixels (dswx_batch_create's default) is no longer needed
 * for speed.  Strides that are not multiples of 8 pixels run the same kernel too (it starts every tile at its first
 * 8-pixel boundary; ABI unchanged, round 5).  Planes at ANY address (int16 planes 2-byte aligned) and, in 'cover' mode, any
 * stride take the same kernel since round 6, through unaligned 16-byte accesses: 0.59 - 0.72 of the HBM peak instead of the
 * aligned layouts' 0.75 - 0.80.
 */
#ifndef DSWX_HIP_H
#define DSWX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSWX_ABI_VERSION 6

enum {
    DSWX_OK = 0,
    DSWX_ERR_ARG = -1,          /* bad argument (NULL, size, non-finite threshold) */
    DSWX_ERR_HIP = -2,          /* a HIP runtime call failed                        */
    DSWX_ERR_NO_DEVICE = -3,    /* no usable gfx950 device                          */
    DSWX_ERR_UNSUPPORTED = -4,  /* e.g. adjacent-to-cloud mode outside 0..2          */
    DSWX_ERR_ALIGN = -5         /* device pointer not aligned for its element type  */
};

/* mask_adjacent_to_cloud_mode, dswx_hls.py:1919-1993 / :1996-2086 */
enum { DSWX_ADJ_MASK = 0, DSWX_ADJ_IGNORE = 1, DSWX_ADJ_COVER = 2 };

/* index of the counters a tile produces, dswx_hls.py:5104-5112 */
enum { DSWX_N_VALID = 0, DSWX_N_CLOUD_AND_VALID = 1, DSWX_N_NOT_OCEAN = 2,
       DSWX_N_COUNTERS = 3 };

/*
 * Everything that parameterises the chain.
 *  - the twelve doubles are HlsThresholds (dswx_hls.py:274-318, defaults
 *    defaults/dswx_hls.yaml:176-212); they must be finite.
 *  - band_fill / fmask_fill: the `image == fill_value` test of
 *    _load_hls_band_from_file (:2195-2209); NaN disables the test for a plane.
 *  - clip_negative_reflectance: FLAG_CLIP_NEGATIVE_REFLECTANCE (:31, :2298).
 *  - aerosol_fmask_lut[k][v] != 0  <=>  Fmask value v is in the runconfig list
 *    for WTR-1 class {0,2,3,4}[k] (:1249-1302, defaults yaml :77-89).
 *  - aerosol_max_nir: AEROSOL_REMAPPING_MAX_NIR (:45-46) = 1000.0.
 *  - collapse_wtr_classes: FLAG_COLLAPSE_WTR_CLASSES (:26); when set, WTR, WTR-1,
 *    WTR-1-AEROSOL and WTR-2 leave in the collapsed form the reference SAVES
 *    (_collapse_wtr_classes :2578-2598, applied at :2688-2689).
 *  - offset_and_scale_inputs: flag_offset_and_scale_inputs (CLI --offset-and-scale-inputs, :2300-2302).  When set,
 *    every clipped reflectance becomes band_scale[k] * (float32(value) - band_offset[k]) -- the `scale_factor` and
 *    `add_offset` of the band's metadata (:2295-2298) -- and the whole chain (indices, five tests, `nir <=
 *    aerosol_max_nir`, `nir > lcmask_nir`) is evaluated in float32 as numpy does on float32 arrays (thresholds rounded
 *    to float32 first: a Python float is a weak scalar against a float32 array).  The fill test stays on the raw
 *    integers.  Same kernels, the five tests in float32 (since round 5: 0.77 of the HBM peak; the production
 *    configuration leaves this off).
 *  - browse_*: the keyword arguments of _compute_browse_array (:3057-3064); the browse
 *    layer is derived from the UNCOLLAPSED WTR (PSW-aggressive is dropped before the
 *    collapse, :3112-3119), which only exists inside the kernel.
 */
typedef struct dswx_params {
    double wigt, awgt, pswt_1_mndwi, pswt_1_nir, pswt_1_swir1, pswt_1_ndvi,
           pswt_2_mndwi, pswt_2_blue, pswt_2_nir, pswt_2_swir1, pswt_2_swir2,
           lcmask_nir;
    double band_fill[6];
    double fmask_fill;
    double aerosol_max_nir;
    int32_t clip_negative_reflectance;
    int32_t mask_adjacent_to_cloud_mode;
    int32_t apply_aerosol_class_remapping;
    int32_t collapse_wtr_classes;
    /* browse layer, _compute_browse_array :3057-3129 (only read when out->browse != NULL) */
    int32_t browse_exclude_psw_aggressive;
    int32_t browse_not_water_to_nodata;
    int32_t browse_cloud_to_nodata;
    int32_t browse_snow_to_nodata;
    int32_t browse_ocean_masked_to_nodata;
    int32_t offset_and_scale_inputs;       /* ABI <= 3: reserved, zero */
    uint8_t aerosol_fmask_lut[4][256];
    double band_scale[6];                  /* only read when offset_and_scale_inputs != 0 */
    double band_offset[6];
} dswx_params_t;

/* Inputs.  band[] order: blue, green, red, nir, swir1, swir2 (raw int16 as read
 * from the HLS files, fill values still in place).  land / shad / ocean are
 * optional (NULL = layer not given): LAND codes :252-264, SHAD 0 = masked
 * (:168-169), OCEAN 0 = ocean (:5243-5245). */
typedef struct dswx_planes_in {
    const int16_t* band[6];
    const uint8_t* fmask;
    const uint8_t* land;
    const uint8_t* shad;
    const uint8_t* ocean;
} dswx_planes_in_t;

/* Outputs; any pointer may be NULL (that layer is then not produced).
 *   diag          UInt16 decimal-digit rendering, nodata 65535  (:4286-4317, :5227)
 *   wtr1          WTR-1 as SAVED, i.e. before aerosol remapping (:5229-5258)
 *   wtr1_aerosol  WTR-1 after _apply_aerosol_class_remapping    (:5260-5266)
 *   wtr2          _apply_landcover_and_shadow_masks             (:5268)
 *   wtr           _apply_cloud_masking                          (:5286)
 *   bwtr          _get_binary_water_layer                       (:5358)
 *   conf          _get_confidence_layer                         (:5368)
 *   cloud         _add_snow_to_cloud_layer                      (:5282)
 *   browse        _compute_browse_array on WTR                  (:5309-5316)
 *   mndwi/ndvi/awesh  float64 spectral indices (:1872-1887); debug planes, they
 *                 select a slower kernel variant.                                   */
typedef struct dswx_planes_out {
    uint16_t* diag;
    uint8_t* wtr1;
    uint8_t* wtr1_aerosol;
    uint8_t* wtr2;
    uint8_t* wtr;
    uint8_t* bwtr;
    uint8_t* conf;
    uint8_t* cloud;
    uint8_t* browse;
    double* mndwi;
    double* ndvi;
    double* awesh;
} dswx_planes_out_t;

/* Geometry of a device-resident batch; tile_stride in pixels, 0 = height*width. */
typedef struct dswx_batch_geom {
    int64_t n_tiles;
    int64_t height;
    int64_t width;
    int64_t tile_stride;
} dswx_batch_geom_t;

typedef struct dswx_ctx dswx_ctx_t;

/* ---- library / context ------------------------------------------------------ */
int dswx_abi_version(void);
const char* dswx_last_error(void);
int dswx_device_count(void);
int dswx_ctx_create(int device, dswx_ctx_t** out);
int dswx_ctx_destroy(dswx_ctx_t* ctx);
/* fills `p` with defaults/dswx_hls.yaml:73-101,176-212 and the constants above */
int dswx_params_default(dswx_params_t* p);

/* ---- the hot path ------------------------------------------------------------ */
/* Host-pointer entry: replaces dswx_hls.py:5089, :5110-5112, :5225-5231,
 * :5245-5249, :5261, :5268, :5282, :5286, :5358, :5368 for a batch of tiles.
 * Stages H2D, runs the fused kernel, stages D2H.  counters: [n_tiles][3] or NULL. */
int dswx_classify_host(dswx_ctx_t* ctx, const dswx_params_t* params,
                       int64_t n_tiles, int64_t height, int64_t width,
                       const dswx_planes_in_t* in, const dswx_planes_out_t* out,
                       int64_t* counters);

/* Device-pointer entry (inputs already resident in HBM): asynchronous on
 * `stream` (a hipStream_t, NULL = the context's stream).  counters: device
 * int64 [n_tiles][3] or NULL; they are zeroed on the stream first.  Mode 'cover'
 * returns DSWX_ERR_UNSUPPORTED here (no geometry): use dswx_classify_device_2d. */
int dswx_classify_device(dswx_ctx_t* ctx, const dswx_params_t* params,
                         int64_t n_tiles, int64_t n_pixels,
                         const dswx_planes_in_t* in, const dswx_planes_out_t* out,
                         int64_t* counters, void* stream);

/* Same, with the tile geometry (n_pixels = height * width).  Required for
 * mask_adjacent_to_cloud_mode 'cover' (_add_snow_to_cloud_layer :2055-2078), whose
 * masked dilations are a 2-D neighbourhood operation: the fused kernel then stops
 * before the snow step and a second, LDS-tiled kernel dilates and finishes CLOUD,
 * WTR, BWTR and CONF.  For 'mask' / 'ignore' it is identical to dswx_classify_device. */
int dswx_classify_device_2d(dswx_ctx_t* ctx, const dswx_params_t* params,
                            int64_t n_tiles, int64_t height, int64_t width,
                            const dswx_planes_in_t* in, const dswx_planes_out_t* out,
                            int64_t* counters, void* stream);

/* The general device entry: dswx_classify_device_2d plus an explicit tile stride. */
int dswx_classify_batch(dswx_ctx_t* ctx, const dswx_params_t* params,
                        const dswx_batch_geom_t* geom, const dswx_planes_in_t* in,
                        const dswx_planes_out_t* out, int64_t* counters, void* stream);

/* generate_interpreted_layer (dswx_hls.py:1687-1707) alone: `n` DIAG values in
 * decimal (0..31 -> class 0..4 per interpreted_dswx_band_dict :97-143; 32 and any
 * other value -> 255), host pointers.  This is the function the reference's unit
 * test exercises (tests/test_dswx_hls_units.py:7-28). */
int dswx_interpret_layer_host(dswx_ctx_t* ctx, const int64_t* diag_decimal, int64_t n,
                              uint8_t* out);

/* Terrain shadow layer: _compute_opera_shadow_layer (dswx_hls.py:4215-4283) followed
 * by the DEM-margin crop (_crop_2d_array_all_sides :4320, used at :5170).
 *   dem       float32 [height][width] including the margin (DEM_MARGIN_IN_PIXELS = 50, :58)
 *   shadow    uint8 [height-2*margin][width-2*margin]; 1 = not shadow, 0 = shadow
 *   sun_vector, sin_azimuth, cos_azimuth: the float64 scalars the reference derives from
 *             the sun angles (:4246-4253, :4276-4277); the caller computes them (the Python
 *             host does it with numpy exactly as the reference), so that they are
 *             bit-identical to the reference's whatever libm is in use
 *   thresholds in degrees (:4279-4281); pixel spacings as :4217 (default 30, 30).
 * float32 / float64 split of the arithmetic: see the kernel comment (numpy >= 2 promotion). */
int dswx_shadow_layer_host(dswx_ctx_t* ctx, const float* dem, int64_t height, int64_t width,
                           int64_t margin, const double sun_vector[3], double sin_azimuth,
                           double cos_azimuth, double min_slope_angle,
                           double max_sun_local_inc_angle, double pixel_spacing_x,
                           double pixel_spacing_y, uint8_t* shadow);
/* The two angle tests pulled back onto the arguments of arccos / arctan (both monotonic):
 *   degrees(arccos(q)) <= max_sun_local_inc_angle  <=>  *inc_q_min <= q <= 1
 *   degrees(arctan(t)) <= min_slope_angle          <=>  t <= *slope_arg_max
 * found by bisection over the doubles with libm.  dswx_shadow_layer_host/_device call this;
 * a caller that wants the boundary of ITS math library (the Python host uses numpy's own
 * arccos / arctan, as the reference does) computes the pair itself and calls the _q forms. */
int dswx_shadow_thresholds(double min_slope_angle, double max_sun_local_inc_angle,
                           double* slope_arg_max, double* inc_q_min);
int dswx_shadow_layer_host_q(dswx_ctx_t* ctx, const float* dem, int64_t height, int64_t width,
                             int64_t margin, const double sun_vector[3], double sin_azimuth,
                             double cos_azimuth, double slope_arg_max, double inc_q_min,
                             double pixel_spacing_x, double pixel_spacing_y, uint8_t* shadow);
int dswx_shadow_layer_device_q(dswx_ctx_t* ctx, const float* dem, int64_t n_tiles,
                               int64_t height, int64_t width, int64_t margin,
                               const double sun_vector[3], double sin_azimuth, double cos_azimuth,
                               double slope_arg_max, double inc_q_min, double pixel_spacing_x,
                               double pixel_spacing_y, uint8_t* shadow, void* stream);
/* float32 variants: the value-based casting of numpy < 2 (the reference pins numpy 1.23.5), under
 * which the float64 sun scalars do not upcast the float32 DEM arrays, so every product, sum,
 * quotient, arccos / arctan / degrees and comparison of dswx_hls.py:4264-4281 is float32 (scalars
 * rounded to float32 first).  Thresholds are float32 values located with float32 arccos / arctan
 * (proteus_amd._capi.shadow_thresholds(..., float32=True)).  The _q forms above reproduce numpy >= 2
 * (NEP 50), where those steps are float64. */
int dswx_shadow_layer_host_q32(dswx_ctx_t* ctx, const float* dem, int64_t height, int64_t width,
                               int64_t margin, const double sun_vector[3], double sin_azimuth,
                               double cos_azimuth, float slope_arg_max, float inc_q_min,
                               double pixel_spacing_x, double pixel_spacing_y, uint8_t* shadow);
int dswx_shadow_layer_device_q32(dswx_ctx_t* ctx, const float* dem, int64_t n_tiles,
                                 int64_t height, int64_t width, int64_t margin,
                                 const double sun_vector[3], double sin_azimuth, double cos_azimuth,
                                 float slope_arg_max, float inc_q_min, double pixel_spacing_x,
                                 double pixel_spacing_y, uint8_t* shadow, void* stream);
/* Device-pointer form for `n_tiles` DEMs of equal size, asynchronous on `stream`. */
int dswx_shadow_layer_device(dswx_ctx_t* ctx, const float* dem, int64_t n_tiles,
                             int64_t height, int64_t width, int64_t margin,
                             const double sun_vector[3], double sin_azimuth, double cos_azimuth,
                             double min_slope_angle, double max_sun_local_inc_angle,
                             double pixel_spacing_x, double pixel_spacing_y, uint8_t* shadow,
                             void* stream);

/* LAND layer, the per-pixel part of create_landcover_mask (dswx_hls.py:994-1115) after
 * the two GDAL warps: `worldcover_up3` is the ESA WorldCover map on the 3x finer grid
 * (uint8 [3*height][3*width]), `copernicus` the CGLS-100m map on the HLS grid (uint8
 * [height][width]).  3x3 sums of WorldCover {80,90,95} (water), 50 (urban), 10 (tree,
 * only where the CGLS class is one of `forest_classes`) -- decimate_by_summation
 * :874-904 -- then the hierarchy of :1058-1103 with `thresholds` = [evergreen,
 * low-intensity developed, high-intensity developed, water] (landcover_threshold_dict
 * :270-271) and year_offset = WorldCover year - 2000.  land: uint8 [height][width]. */
int dswx_landcover_mask_host(dswx_ctx_t* ctx, const uint8_t* worldcover_up3,
                             const uint8_t* copernicus, int64_t height, int64_t width,
                             const int32_t* forest_classes, int32_t n_forest_classes,
                             const int32_t thresholds[4], int32_t year_offset, uint8_t* land);

/* Device-pointer form for `n_tiles` map pairs of equal size ([n_tiles][3H][3W] and
 * [n_tiles][H][W]), asynchronous on `stream`. */
int dswx_landcover_mask_device(dswx_ctx_t* ctx, const uint8_t* worldcover_up3,
                               const uint8_t* copernicus, int64_t n_tiles, int64_t height,
                               int64_t width, const int32_t* forest_classes,
                               int32_t n_forest_classes, const int32_t thresholds[4],
                               int32_t year_offset, uint8_t* land, void* stream);

/* The two layer kernels writing straight into the SHAD / LAND planes of a resident batch: as the _device forms, with the
 * distance between the output rasters of consecutive tiles given explicitly (bytes = pixels; 0 = packed rasters;
 * dswx_batch_geom_t.tile_stride of the batch whose plane `shadow` / `land` is).  dswx_shadow_layer_batch takes the
 * pulled-back thresholds like the _q / _q32 forms (float32_arithmetic != 0: slope_arg_max and inc_q_min hold float32
 * values, numpy < 2 promotion). */
int dswx_shadow_layer_batch(dswx_ctx_t* ctx, const float* dem, int64_t n_tiles, int64_t height,
                            int64_t width, int64_t margin, const double sun_vector[3], double sin_azimuth,
                            double cos_azimuth, double slope_arg_max, double inc_q_min,
                            int32_t float32_arithmetic, double pixel_spacing_x, double pixel_spacing_y,
                            uint8_t* shadow, int64_t shadow_tile_stride, void* stream);
int dswx_landcover_mask_batch(dswx_ctx_t* ctx, const uint8_t* worldcover_up3, const uint8_t* copernicus,
                              int64_t n_tiles, int64_t height, int64_t width,
                              const int32_t* forest_classes, int32_t n_forest_classes,
                              const int32_t thresholds[4], int32_t year_offset, uint8_t* land,
                              int64_t land_tile_stride, void* stream);

/* Deterministic synthetic HLS tiles written straight into HBM (SURVEY.md §8d;
 * same integer recipe as proteus_amd/synth.py).  Fills in->band[0..5], in->fmask
 * and whichever of land/shad/ocean is non-NULL for tiles tile0..tile0+n_tiles-1. */
int dswx_synth_fill(dswx_ctx_t* ctx, uint64_t seed, int64_t tile0, int64_t n_tiles,
                    int64_t height, int64_t width, const dswx_planes_in_t* in,
                    void* stream);

/* dswx_synth_fill with an explicit tile stride. */
int dswx_synth_batch(dswx_ctx_t* ctx, uint64_t seed, int64_t tile0,
                     const dswx_batch_geom_t* geom, const dswx_planes_in_t* in, void* stream);

/* ---- resident batches: allocation and placement of the planes (ABI v4) ---------
 * The seam at dswx_hls.py:5225-5286 works on arrays the caller already holds; a service that keeps a
 * batch of tiles resident in HBM (bench.py, proteus_amd/batch.py) needs them allocated -- and on MI355X
 * WHERE the seven output planes lie decides between 0.71 and 0.80 of the HBM peak for the same launch
 * (the rate follows the position of the write streams in the address space, with
 * a 32-GiB structure; round 3 looked for a layout RULE and found none that holds from one process to
 * the next -- profiles/r03_placement_rule_trials.json).  dswx_batch_create allocates the planes of a
 * batch (one hipMalloc, packed, unless a flag says otherwise); dswx_batch_place_slide (what bench.py uses)
 * and dswx_batch_place_search are the opt-in measured placements, so the placed rate is available to
 * every caller of this ABI, not to a Python helper only. */
typedef struct dswx_batch dswx_batch_t;

/* plane indices of dswx_batch_layout_t */
enum {
    DSWX_PLANE_BAND0 = 0,      /* blue, green, red, nir, swir1, swir2 = 0..5 (int16) */
    DSWX_PLANE_FMASK = 6,
    DSWX_PLANE_LAND = 7, DSWX_PLANE_SHAD = 8, DSWX_PLANE_OCEAN = 9,          /* DSWX_BATCH_MASKS */
    DSWX_PLANE_DIAG = 10,      /* uint16 */
    DSWX_PLANE_WTR1 = 11, DSWX_PLANE_WTR1_AEROSOL = 12, DSWX_PLANE_WTR2 = 13, DSWX_PLANE_WTR = 14,
    DSWX_PLANE_BWTR = 15, DSWX_PLANE_CONF = 16, DSWX_PLANE_CLOUD = 17, DSWX_PLANE_BROWSE = 18,
    DSWX_PLANE_COUNTERS = 19,  /* int64 [n_tiles][3] */
    DSWX_BATCH_MAX_PLANES = 20
};

/* flags of dswx_batch_create / dswx_batch_layout */
enum {
    DSWX_BATCH_MASKS = 1 << 0,            /* LAND, SHAD, OCEAN input planes (BASELINE config 5) */
    DSWX_BATCH_WTR1_AEROSOL = 1 << 1,     /* optional output planes */
    DSWX_BATCH_BROWSE = 1 << 2,
    /* layout: default = ONE allocation, inputs then outputs back to back (what a plain caller does) */
    DSWX_BATCH_SEPARATE_OUTPUTS = 1 << 10, /* one allocation for the inputs, one per output plane: what
                                              dswx_batch_place_search needs */
    DSWX_BATCH_SLIDING_OUTPUTS = 1 << 11   /* the output planes packed in a range of the virtual address space that
                                              is backed chunk by chunk (HIP virtual memory management): what
                                              dswx_batch_place_slide needs */
};

typedef struct dswx_batch_layout {
    int64_t tile_stride;                              /* pixels, as resolved (see dswx_batch_create) */
    uint64_t arena_bytes;                             /* the single allocation (SEPARATE_OUTPUTS: the inputs') */
    uint64_t plane_bytes[DSWX_BATCH_MAX_PLANES];      /* 0 = plane absent */
    uint64_t plane_offset[DSWX_BATCH_MAX_PLANES];     /* inside the arena (SEPARATE_OUTPUTS: 0 for outputs;
                                                         SLIDING_OUTPUTS: inside the output region) */
    uint64_t write_span_bytes;                        /* first byte of the first to last byte of the last
                                                         output plane (0 with SEPARATE_OUTPUTS) */
} dswx_batch_layout_t;

typedef struct dswx_batch_info {
    dswx_batch_geom_t geom;           /* tile_stride resolved */
    uint32_t flags;
    int32_t n_allocations;
    uint64_t bytes_allocated;         /* HBM the batch holds */
    /* record of the last dswx_batch_place_search (zeros if none ran) */
    int32_t search_candidates;
    int32_t search_probes;            /* probe measurements taken (each `launches` launches) */
    float first_come_launch_ms;       /* the planes as first allocated ... */
    float kept_launch_ms;             /* ... and as kept, timed back to back at the end of the search */
    /* ABI v5: address space (DSWX_BATCH_SLIDING_OUTPUTS; see dswx_batch_va_budget) */
    uint64_t va_reserved_bytes;       /* the reservation this batch's output range lives in (0: no range) */
    uint64_t va_retired_bytes;        /* process-wide: address space of dropped ranges, reserved and empty for good */
    uint64_t va_budget_bytes;         /* process-wide limit on reserved + retired address space */
    uint64_t va_pooled_bytes;         /* process-wide: physical chunks of dropped ranges kept for later ranges */
    char note[256];                   /* "" or why a sliding batch was allocated packed / why the last
                                         dswx_batch_place_slide left the planes where they were */
} dswx_batch_info_t;

/* n_tiles of dswx_batch_classify: every resident tile (ABI v5; up to v4 this was spelled 0, which made an empty last
 * chunk of a walk re-classify the whole batch -- 0 is now what it says: no tiles, no work) */
#define DSWX_BATCH_ALL_TILES (-1)

/* The layout rule alone (pure function, no device needed): where dswx_batch_create would put every
 * plane.  geom->tile_stride 0 = height*width rounded up to a multiple of 256 pixels (every tile of
 * every plane then starts on a 256-byte boundary: the fast kernel, see the top of this file); any
 * other value is taken as given (>= height*width).  Plane offsets are 256-byte aligned. */
int dswx_batch_layout(const dswx_batch_geom_t* geom, uint32_t flags, dswx_batch_layout_t* out);

/* DSWX_BATCH_SLIDING_OUTPUTS on a device without HIP virtual memory management: DSWX_ERR_UNSUPPORTED.  If the address
 * range cannot be had -- hipMemAddressReserve refuses, or the library's address-space budget is spent
 * (dswx_batch_va_budget) -- the batch is allocated PACKED (as without the flag) and dswx_batch_info_t.note says why;
 * dswx_batch_info_t.flags shows the layout in effect.  dswx_batch_place_slide on such a batch times the planes where they
 * are and succeeds (search_probes 0). */
int dswx_batch_create(dswx_ctx_t* ctx, const dswx_batch_geom_t* geom, uint32_t flags,
                      dswx_batch_t** out);
/* Frees every allocation of the batch.  A batch must not be USED after its context is destroyed; destroying it
 * afterwards is allowed. */
int dswx_batch_destroy(dswx_batch_t* batch);
/* Address space and memory of the sliding ranges (ABI v5).  Two properties of HIP virtual memory management on ROCm 7.2 /
 * gfx950 (plain-HIP reproducers: tools/vmm_reuse_repro.hip, tools/lab/vmm_meminfo.hip):
 *   (1) an address that a kernel has accessed through one mapping must never be mapped onto other physical memory -- the
 *       kernel's translation stays stale and its stores go to the released memory;
 *   (2) the physical memory of a chunk that was ever mapped returns to the device only when the address RESERVATION it was
 *       mapped in is freed.
 * By default the library therefore RETIRES every sliding range it drops: the addresses stay reserved, empty, for the life
 * of the process (`retired_bytes`; `live_bytes` = reserved by ranges in use), and the physical chunks go into a process-wide
 * POOL (`pooled_bytes`) from which later batches and placements of the same chunk size are built before new memory is
 * created -- nothing is ever exposed to reuse, and a service that re-creates or re-places batches reuses its own memory.
 * dswx_batch_pool_trim() gives the pooled memory back to the device: it releases the chunks, frees every retired
 * reservation -- which is what returns the memory -- and reserves the same addresses again at once, empty.  Between those
 * two calls the addresses are up for grabs by other threads of the process: call it when none of them allocates (a range
 * lost that way is counted in `loose_bytes` and is out of the library's control).  Batches may be LIVE during a trim,
 * placed ones included: a kept placement's chunks were once mapped in the (now retired) wide range whose reservation the trim
 * frees, and that was measured to be harmless (tests/test_gpu_parity.py::test_pool_trim_while_a_placed_batch_is_live;
 * profiles/r05_trim_live_probe.json).  The library never trims by itself, not even when an allocation fails: the error
 * text of dswx_batch_create then says how many bytes the pool holds.  Chunk sizes are powers of two (2 MiB ... 1 GiB), so
 * batches of different geometry share the pool.
 * Address space is consumed for good: 100 - 160 GiB per placed batch at 256 tiles of 3660 x 3660 (the first-come range, the
 * wide range and, when the batch goes, the range of the kept chunks), so the default BUDGET of 64 TiB (half of the 47-bit
 * space) lasts 400 - 650 placements.  When live + retired + a new request would pass the budget the library reserves no
 * more: sliding batches fall back as described at dswx_batch_create.
 * new_budget_bytes 0 = leave the budget as it is; any output pointer may be NULL.  Process-wide, thread-safe. */
int dswx_batch_va_budget(uint64_t new_budget_bytes, uint64_t* budget_bytes, uint64_t* live_bytes,
                         uint64_t* retired_bytes, uint64_t* loose_bytes, uint64_t* pooled_bytes);
int dswx_batch_pool_trim(uint64_t* released_bytes);
/* Device pointers of the planes (absent planes NULL), the resolved geometry and the counters array
 * ([n_tiles][3] int64); any output argument may be NULL.  Hand them to dswx_classify_batch /
 * dswx_synth_batch, or use the two conveniences below. */
int dswx_batch_planes(const dswx_batch_t* batch, dswx_batch_geom_t* geom, dswx_planes_in_t* in,
                      dswx_planes_out_t* out, int64_t** counters);
int dswx_batch_info(const dswx_batch_t* batch, dswx_batch_info_t* info);
/* dswx_classify_batch over the first `n_tiles` resident tiles (DSWX_BATCH_ALL_TILES = all, 0 = none), counters
 * included. */
int dswx_batch_classify(dswx_batch_t* batch, const dswx_params_t* params, int64_t n_tiles,
                        void* stream);
/* dswx_synth_batch into the resident input planes: tiles tile0 .. tile0 + n_tiles - 1. */
int dswx_batch_synth(dswx_batch_t* batch, uint64_t seed, int64_t tile0, void* stream);
/* Opt-in, measured placement (DSWX_BATCH_SEPARATE_OUTPUTS batches whose inputs are resident): beside
 * every output plane `candidates - 1` spare allocations are made (as many sets as fit while
 * `keep_free_bytes` of device memory stay free), and one pass of coordinate descent binds each plane in
 * turn (DIAG first) to the candidate under which `launches` launches of the real kernel run fastest;
 * the chosen and the first-come planes are then timed back to back and the better set is kept.  The
 * spares are freed.  Output pointers change: call dswx_batch_planes again.  Synchronous. */
int dswx_batch_place_search(dswx_batch_t* batch, const dswx_params_t* params, int32_t candidates,
                            int32_t launches, uint64_t keep_free_bytes);
/* The cheaper measured placement (DSWX_BATCH_SLIDING_OUTPUTS batches whose inputs are resident).  The rate follows
 * the POSITION of the packed output region in the address space with a structure of ~32 GiB, so
 * one dimension is enough: a range `slack_bytes` longer than the output planes is mapped beside the current one
 * (bounded so that `keep_free_bytes` of device memory stay free), the kernel is timed (`launches` launches) with the
 * output region at offsets 0, step_bytes, 2 step_bytes, ... of it, the best position and the first-come range are
 * then timed back to back, and the better one is kept: the chunks (physical memory) under the chosen position move into a
 * fresh address range of their own, the wide range is dropped (its other chunks go to the library's pool, see
 * dswx_batch_va_budget / dswx_batch_pool_trim); the moved planes are timed once more and kept only if they still beat the
 * first-come range.  After the packed positions `spread_gaps` more
 * candidates are tried: the planes spread over the range with equal gaps of 1/spread_gaps ... 1 x the largest gap that
 * fits (0 = packed positions only).  `refine_passes` passes of refinement follow: from the best candidate, every
 * plane in turn (DIAG first) tries the other free places of the range on a grid of 2 step_bytes and keeps the best
 * (the per-plane freedom of dswx_batch_place_search without its spare allocations; ~12 probes per plane and pass).
 * slack / step + 1 + spread_gaps probes without refinement (25 + 4 at 48 GiB / 2 GiB:
 * about 1 s for 256 tiles) and slack_bytes of transient memory, against 185 probes and five spare sets of planes
 * for dswx_batch_place_search.  Output pointers change: call dswx_batch_planes again.  Synchronous.
 * (Every placement retires the address ranges it drops: see dswx_batch_va_budget.  When the
 * wide range cannot be reserved or mapped the planes stay where they are, the call succeeds and dswx_batch_info_t.note
 * says why.) */
int dswx_batch_place_slide(dswx_batch_t* batch, const dswx_params_t* params, uint64_t slack_bytes,
                           uint64_t step_bytes, int32_t spread_gaps, int32_t refine_passes,
                           int32_t launches, uint64_t keep_free_bytes);

/* ---- the raster formats either side of the path (ABI v6) --------------------------------
 * The reference reads and saves every raster through GDAL.  Reading a band file is inflate -> inverse predictor ->
 * blocks into the raster (`ReadAsArray` in _load_hls_band_from_file, dswx_hls.py:2136-2302); saving a layer is
 * `save_as_cog` (core.py:7-91): NEAREST overviews 4 / 16 / 64 / 128 for the integer layers (:37-46), 512 x 512 blocks,
 * PREDICTOR=2 (integers) / 3 (floating point), DEFLATE (:60-75); the RGB composites are
 * scale * (float32(band) - offset) with NaN on invalid pixels (_save_output_rgb_file, dswx_hls.py:3013-3036).
 * Compression stays on the host (include/dswx_codec.h); the byte shuffling between an inflated block and the
 * classifier's planes, and between its layers and the blocks to deflate, runs on the device with these entries.
 * All pointers are DEVICE pointers; asynchronous on `stream` (NULL = the context's stream). */
#define DSWX_COG_MAX_LEVELS 8
typedef struct dswx_cog_layout {
    int32_t n_levels;                                /* the full-resolution image + the overview levels */
    int32_t tile;
    int32_t factor[DSWX_COG_MAX_LEVELS];             /* 1, then the factors that produce a level */
    int64_t height[DSWX_COG_MAX_LEVELS];             /* ceil(N / factor), what GDAL makes an overview */
    int64_t width[DSWX_COG_MAX_LEVELS];
    int32_t blocks_down[DSWX_COG_MAX_LEVELS];
    int32_t blocks_across[DSWX_COG_MAX_LEVELS];
    uint64_t offset_bytes[DSWX_COG_MAX_LEVELS];      /* of the level's first block in the blocked buffer */
    uint64_t total_bytes;
} dswx_cog_layout_t;
/* Where dswx_cog_blocks_device puts what (pure function, no device needed).  A factor of 1, or any factor on a 1 x 1
 * raster, produces no level (the host writer's rule). */
int dswx_cog_layout(int64_t height, int64_t width, int32_t elem_bytes, int32_t tile, const int32_t* factors,
                    int32_t n_factors, dswx_cog_layout_t* out);
/* One plane [height][width] -> `blocks`: for every level of dswx_cog_layout, its tile x tile blocks in row-major block
 * order, edge blocks zero-padded, each block row predictor-encoded and little endian -- exactly the bytes the TIFF writer
 * hands to DEFLATE.  elem_bytes 1 / 2: integer samples, predictor 1 (none) or 2 (horizontal differencing in the sample's
 * width), overview levels by GDAL's NEAREST rule (src = min(int(0.5 + dst * N / N_ovr), N - 1), restated from
 * GDALResampleChunk_Near: GDAL is not in the reference tree, parity of the rule itself is unpinned).  elem_bytes 4:
 * Float32 with predictor 3 (TIFF Technical Note 3), one level per call: the reference's overviews of a Float32 layer are
 * CUBICSPLINE -- build each level with dswx_convolve_axis_device (two passes) and pass it here on its own. */
int dswx_cog_blocks_device(dswx_ctx_t* ctx, const void* plane, int32_t elem_bytes, int64_t height, int64_t width,
                           int32_t tile, const int32_t* factors, int32_t n_factors, int32_t predictor, void* blocks,
                           void* stream);
/* The inverse for a file being read: `blocks` = every block of one plane of an image, inflated, in block order, each
 * block_height x block_width samples (tiles; or strips: block_width = width, the short last strip's slot padded), native
 * byte order; elem_bytes 1 / 2 / 4 with predictor 1 (none) or 2 (running sum in the sample's width), or elem_bytes 4 with
 * predictor 3 (Float32, TIFF Technical Note 3: what GDAL writes for a Float32 DEM with PREDICTOR=3).  -> plane [height][width]. */
int dswx_untile_device(dswx_ctx_t* ctx, const void* blocks, int32_t elem_bytes, int64_t height, int64_t width,
                       int32_t block_width, int32_t block_height, int32_t predictor, void* plane, void* stream);
/* One separable pass of the CUBICSPLINE overview convolution `save_as_cog` asks GDAL for on non-integer layers
 * (core.py:41-46; restated from GDAL's GDALResampleChunk_Convolution in proteus_amd/geotiff.py: _convolve_axis -- GDAL is not
 * in the reference tree, last-ulp agreement of the float results is unpinned): for every line r < n_lines and output
 * position j < n_out, dst = sum_k src[r, clamp(first[j] + k, 0, n_in - 1)] * weights[k * n_out + j], normalised over the taps whose
 * sample is not NaN; NaN if none is left.  Accumulation in float64 in tap order.  Strides in elements, so that the same entry
 * does the horizontal pass (lines = rows) and the vertical one (lines = columns); src / dst float32 or float64; `first`
 * (int32 [n_out]) and `weights` (float64 [taps][n_out]) are DEVICE arrays the host prepares. */
int dswx_convolve_axis_device(dswx_ctx_t* ctx, const void* src, int32_t src_is_f64, int64_t n_lines, int64_t n_in,
                              int64_t src_line_stride, int64_t src_elem_stride, int64_t n_out, int32_t taps,
                              const int32_t* first, const double* weights, void* dst, int32_t dst_is_f64,
                              int64_t dst_line_stride, int64_t dst_elem_stride, void* stream);
/* Rows of `width_bytes` bytes from one device raster to another (hipMemcpy2DAsync, device to device): the crop of a DEM
 * with its margin to the product grid (_crop_2d_array_all_sides, dswx_hls.py:4320) without a trip to the host. */
int dswx_copy_2d_device(dswx_ctx_t* ctx, void* dst, size_t dst_pitch_bytes, const void* src, size_t src_pitch_bytes,
                        size_t width_bytes, size_t height, void* stream);
/* out[c][i] = scale[c] * (float32(band_c[i], clipped to >= 1 if clip_negative_reflectance) - offset[c]) in float32, NaN
 * where diag[i] == 65535 (diag may be NULL: no masking); out = float [3][n_pixels]. */
int dswx_rgb_planes_device(dswx_ctx_t* ctx, const int16_t* red, const int16_t* green, const int16_t* blue,
                           const uint16_t* diag, int64_t n_pixels, const double scale[3], const double offset[3],
                           int32_t clip_negative_reflectance, float* out, void* stream);
/* A plane as GDAL stores it in a Byte band (save_dswx_product creates all ten bands of the multi-band file as GDT_Byte,
 * dswx_hls.py:2663-2666, so DIAG and DEM saturate there): src_kind 1 uint16 / 2 int16: clamped to 0 .. 255; 3 float32:
 * NaN -> 0, clamped, rounded half up (GDALCopyWords; GDAL is not in the reference tree: the rule is GDAL's documented
 * conversion, unpinned by execution). */
int dswx_to_byte_device(dswx_ctx_t* ctx, const void* src, int32_t src_kind, int64_t n, uint8_t* dst, void* stream);
/* dst[i][j] = src[rows[i]][cols[j]] (rows / cols: DEVICE int32 arrays of source indices, which the caller guarantees to lie
 * inside src_height x src_width): the nearest-neighbour resampling of the browse image (geotiff2png after
 * _compute_browse_array, dswx_hls.py:5335-5349; GDAL RasterIO's pick src = floor((dst + 0.5) * N_src / N_dst), which the host
 * evaluates -- proteus_amd/geotiff.py: resample_nearest) on a plane that stays in HBM; only the small image crosses PCIe. */
int dswx_gather_2d_device(dswx_ctx_t* ctx, const void* src, int32_t elem_bytes, int64_t src_height, int64_t src_width,
                          const int32_t* rows, int32_t n_rows, const int32_t* cols, int32_t n_cols, void* dst, void* stream);

/* ---- device plumbing for hosts without another HIP binding ------------------- */
int dswx_device_malloc(dswx_ctx_t* ctx, size_t bytes, void** out);
int dswx_device_free(dswx_ctx_t* ctx, void* ptr);
/* Page-locked host memory.  dswx_classify_host() recognises buffers allocated HERE and, when
 * EVERY plane lies inside such a buffer, works on them in place: the host planes are mapped into
 * the device's address space and the kernels read the inputs and write the layers across PCIe
 * themselves (zero copy, both directions at once, every mode) instead of the synchronous
 * copy-compute-copy sequence.  Any other memory -- pageable, or page-locked by the caller with
 * hipHostRegister, whose pages need not be resident -- is copied.  There is no reference
 * counterpart (numpy arrays are pageable); results are identical. */
int dswx_host_alloc(dswx_ctx_t* ctx, size_t bytes, void** out);
/* ctx may be NULL here (a span may outlive the context it was allocated through). */
int dswx_host_free(dswx_ctx_t* ctx, void* ptr);
int dswx_memcpy_h2d(dswx_ctx_t* ctx, void* dst, const void* src, size_t bytes);
int dswx_memcpy_d2h(dswx_ctx_t* ctx, void* dst, const void* src, size_t bytes);
int dswx_memset_d(dswx_ctx_t* ctx, void* dst, int value, size_t bytes);      /* complete on return, like the two copies above */
/* asynchronous on `stream` (NULL = the context's stream); the host side must be page-locked (dswx_host_alloc) for the
 * copy to overlap with anything */
int dswx_memcpy_h2d_async(dswx_ctx_t* ctx, void* dst, const void* src, size_t bytes, void* stream);
int dswx_memcpy_d2h_async(dswx_ctx_t* ctx, void* dst, const void* src, size_t bytes, void* stream);
int dswx_stream_synchronize(dswx_ctx_t* ctx, void* stream);
/* HIP events on the stream the kernels run on (timing for bench.py) */
int dswx_event_create(dswx_ctx_t* ctx, void** out);
int dswx_event_destroy(dswx_ctx_t* ctx, void* event);
int dswx_event_record(dswx_ctx_t* ctx, void* event, void* stream);
int dswx_event_elapsed_ms(dswx_ctx_t* ctx, void* start, void* stop, float* ms);

/* Name and launch geometry of the kernel the last dswx_classify_* call on this
 * context selected (for profiles / DESIGN.md): writes a NUL-terminated string. */
int dswx_last_kernel_info(dswx_ctx_t* ctx, char* buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* DSWX_HIP_H */
