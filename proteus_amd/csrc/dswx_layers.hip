// dswx_layers.hip -- the layers either side of the fused classifier and the
// synthetic-tile generator: generate_interpreted_layer on its own, the terrain shadow layer,
// the LAND 3x3 aggregation, dswx_synth_*; kernels and their C-ABI entry points.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <limits>

#include "dswx_host.h"

// ------------------------------------------------------------------------------
// generate_interpreted_layer (:1687-1707) on its own: DIAG in decimal (any integer,
// as the reference's unit test feeds it) -> WTR-1 class; 32 and anything outside
// the table -> 255.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dswx_interpret_v1(const long long* __restrict__ diag,
                                                         uint8_t* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long d = diag[i];
    uint32_t cls = 255u;
    if (d >= 0 && d < 32) {
        const uint32_t k = (uint32_t)d;
        cls = ((CLS_B0 >> k) & 1u) | (((CLS_B1 >> k) & 1u) << 1) | (((CLS_B2 >> k) & 1u) << 2);
    }
    out[i] = (uint8_t)cls;
}

// ------------------------------------------------------------------------------
// Terrain shadow layer (row f1): _compute_opera_shadow_layer :4215-4283 followed by
// the margin crop of :4320 / :5170.  One thread per OUTPUT pixel.
//
// Arithmetic types follow what numpy >= 2 (NEP 50) gives the reference expressions on a
// float32 DEM: np.gradient, the division by the pixel spacing, the squares, their sum,
// `+ 1` and the sqrt stay float32; the products with the float64 sun-vector scalars,
// the quotient, arccos / arctan / degrees and the comparisons are float64.  (Under the
// numpy 1.23.5 the reference pins, value-based casting keeps those float32 as well:
// borderline pixels can differ between the two -- SURVEY.md §7.)  Built without
// fp contraction; hipcc's float32 division and sqrt are correctly rounded.
//
// No transcendental runs on the device: arccos and arctan are monotonic, so the host pulls
// the two angle thresholds back onto their arguments by bisection over the doubles WITH THE
// LIBRARY THE REFERENCE WOULD USE (the Python host: numpy's own arccos / arctan loops;
// dswx_shadow_thresholds: libm) and the kernel compares the arguments.  That is both
// cheaper (the kernel becomes HBM-bound) and closer to the reference than a device libm:
// whichever way numpy rounds arccos at the threshold, the boundary moves with it.
// ------------------------------------------------------------------------------
struct ShadowArgs {
    const float* dem;        // [H][W], with margin
    uint8_t* shadow;         // [H - 2*margin][W - 2*margin]; 1 = not shadow, 0 = shadow
    long long height, width, margin;
    long long out_stride;    // bytes between the shadow rasters of consecutive tiles ((H - 2 margin) (W - 2 margin) unless a batch says otherwise)
    float spacing_x, neg_abs_spacing_y;
    double sun[3];           // target-to-sun unit vector (x, y, z)
    double sin_az, cos_az;
    // the two angle tests, pulled back through the (monotonic) arccos / arctan onto their
    // arguments by the host (dswx_shadow_thresholds or the caller's own numpy):
    //   degrees(arccos(q)) <= max_sun_local_inc_angle  <=>  inc_q_min <= q <= 1
    //   degrees(arctan(t)) <= min_slope_angle          <=>  t <= slope_arg_max
    double inc_q_min, slope_arg_max;
    // dswx_shadow_v3: block row = by_first + blockIdx.y * by_step (0 / 1 in production; the lab's two-pass launch -- even
    // block rows of every tile, then the odd ones -- puts > 256 MiB of traffic between the two reads of every halo row,
    // which separates what the Infinity Cache absorbs from what reaches HBM: profiles/r06_next_rows_pmc.json)
    int by_first, by_step;
};

// The arithmetic of one pixel exactly as the reference orders it, given the two finite differences
// and their np.gradient scale factors (0.5 inside, 1 on a border).
// F32: the value-based casting of numpy < 2 (what the reference pins: numpy 1.23.5), where the
// float64 sun scalars do NOT upcast the float32 arrays -- every product, sum, quotient, arccos,
// arctan, degrees and comparison of :4264-4281 is float32, the scalars rounded to float32 first.
template <bool F32>
__device__ __forceinline__ bool shadow_px_exact(const ShadowArgs& a, float dx, float sx, float dy, float sy) {
    const float gx = dx * sx, gy = dy * sy;
    const float n0 = -gx / a.spacing_x;
    const float n1 = -gy / a.neg_abs_spacing_y;
    const float norm = sqrtf(n0 * n0 + n1 * n1 + 1.0f);
    bool low_inc, backslope;
    if (F32) {
        const float qf = (n0 * (float)a.sun[0] + n1 * (float)a.sun[1] + (float)a.sun[2]) / norm;
        const float tf = n0 * (float)a.sin_az + n1 * (float)a.cos_az;
        low_inc = (qf >= (float)a.inc_q_min) & (qf <= 1.0f);      // thresholds are float32 values here
        backslope = tf <= (float)a.slope_arg_max;
    } else {
        const double dot = (double)n0 * a.sun[0] + (double)n1 * a.sun[1] + a.sun[2];
        const double q = dot / (double)norm;                      // arccos argument (NaN-safe compares below)
        const double t = (double)n0 * a.sin_az + (double)n1 * a.cos_az;   // arctan argument
        low_inc = (q >= a.inc_q_min) & (q <= 1.0);                // arccos(q > 1) is NaN: the test fails
        backslope = t <= a.slope_arg_max;
    }
    return low_inc | !backslope;
}

// General form: one thread per output pixel, any margin / width / alignment (also the borders of a
// margin-free call, where np.gradient falls back to one-sided differences).
template <bool F32>
__global__ __launch_bounds__(256) void dswx_shadow_v2(const ShadowArgs a) {
    const int W = (int)a.width, H = (int)a.height, margin = (int)a.margin;
    const int ow = W - 2 * margin, oh = H - 2 * margin;
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= ow || oy >= oh) return;
    const int x = ox + margin, y = oy + margin;
    const float* __restrict__ d = a.dem + (size_t)blockIdx.z * (size_t)H * (size_t)W;
    // np.gradient, edge_order 1, unit spacing: central differences inside ((f[i+1] - f[i-1]) / 2,
    // and x / 2 == x * 0.5f exactly), one-sided first differences at the borders; branch-free
    const int c = y * W + x;                       // H * W < 2^31 is checked on the host
    const bool x_in = (x > 0) & (x < W - 1), y_in = (y > 0) & (y < H - 1);
    const float dx = d[x < W - 1 ? c + 1 : c] - d[x > 0 ? c - 1 : c];
    const float dy = d[y < H - 1 ? c + W : c] - d[y > 0 ? c - W : c];
    const bool v = shadow_px_exact<F32>(a, dx, x_in ? 0.5f : 1.0f, dy, y_in ? 0.5f : 1.0f);
    a.shadow[(size_t)blockIdx.z * (size_t)a.out_stride + (size_t)(oy * ow + ox)] = v ? 1 : 0;
}

// ------------------------------------------------------------------------------
// Production form (margin >= 2, output at least 4 pixels wide: every output pixel and the two columns beside a quad
// are interior): FOUR output pixels per thread, 8-byte DEM loads, one dword store, and a floating-point FILTER in
// front of the exact arithmetic.  Up to round 5 it also asked for an even margin, an even DEM width and an output
// width that is a multiple of 4 (8-byte aligned loads, dword-aligned stores); since round 6 the loads and the store
// are UNALIGNED global accesses (gfx950 does them in hardware; same instructions, the reference's 3760 x 3760 / margin 50
// geometry is aligned anyway) and the last quad of a row starts at ow - 4, overlapping its neighbour with identical
// values -- so any margin >= 2, any width, any alignment of the buffers takes this kernel (the
// general kernel below, 0.24 of the HBM peak and 2.0 x the algorithmic traffic, keeps margins 0 / 1 and tiny rasters).
//
// The exact chain costs ~106 VALU per pixel (two correctly rounded float32 divisions, a correctly
// rounded sqrtf, a float64 division) and made the one-pixel kernel issue-bound at 0.2 of the HBM
// rate.  But the result is two threshold tests, and all but a handful of pixels sit far from both
// thresholds.  So every pixel is first evaluated approximately in packed float32 (reciprocal
// multiplies, v_rsq_f32: ~25 VALU) with a rigorous error bound:
//     |q~ - q_ref| <= K  = 2^-18 (|sun_x| + |sun_y| + |sun_z|)           (q = arccos argument)
//     |t~ - t_ref| <= Et = 2^-19 (|n0 sin_az| + |n1 cos_az|) (+ 2^-120)  (t = arctan argument)
// (derivation in DESIGN.md section 3: both chains are within ~20 float32 half-ulps of the real-number
// value, the bound has 3x slack, thresholds are rounded outwards to float32 on the host).  A test
// whose approximate value clears its threshold by more than the bound is DECIDED; otherwise -- or
// when anything is non-finite / huge -- the pixel is recomputed with shadow_px_exact, the very
// arithmetic of the general kernel.  Bit-exact by construction, and the uncertain band is ~4e-6 wide.
// ------------------------------------------------------------------------------
struct ShadowFilter {
    float inv_x, inv_y;                  // RN(1 / (-2 sx)), RN(1 / (2 |sy|)):  n0 ~ (d[x+1] - d[x-1]) * inv_x
    float s0, s1, s2, sin_az, cos_az;    // the five sun scalars rounded to float32
    float q_c, q_h_in, q_h_out;          // u = q~ - q_c:  low_inc decided TRUE if |u| <= q_h_in, FALSE if |u| > q_h_out
    float t_c;                           // v = t~ - t_c:  backslope decided if |v| > e (TRUE if v < 0, FALSE if v > 0)
    float e_rel, e_abs;                  // e = e_rel sqrt(S) + e_abs   (sqrt(S) >= |n0 sin| + |n1 cos| up to |(sin, cos)|)
    float et_rel;                        // t_tiny only: e = et_rel (|n0 sin| + |n1 cos|) + (2^-120 unless both differences are 0)
    int t_tiny;                          // |slope_arg_max| < 2^-100: the threshold is too close to 0 for a relative bound
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x2 __attribute__((aligned(4))) f32x2_u;          // 8 bytes at any float boundary
typedef uint32_t __attribute__((aligned(1))) u32_u;          // 4 bytes anywhere

// One quad (four horizontally adjacent output pixels) from the three DEM rows around it: c0..c3 = centre
// row d[x-2 .. x+5], u0 u1 / b0 b1 = rows above / below d[x .. x+3].  Returns the four 0 / 1 bytes.
template <bool F32, bool TINY>
__device__ __forceinline__ uint32_t shadow_quad(const ShadowArgs& a, const ShadowFilter& f, f32x2 c0, f32x2 c1, f32x2 c2,
                                                f32x2 c3, f32x2 u0, f32x2 u1, f32x2 b0, f32x2 b1) {
    // finite differences of the four pixels as two packed pairs (every output pixel is interior)
    const f32x2 dx[2] = {f32x2{c1.y, c2.x} - f32x2{c0.y, c1.x}, f32x2{c2.y, c3.x} - f32x2{c1.y, c2.x}};
    const f32x2 dy[2] = {b0 - u0, b1 - u1};
    uint32_t out = 0, unsure = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2 n0 = dx[p] * f.inv_x, n1 = dy[p] * f.inv_y;
        const f32x2 S = __builtin_elementwise_fma(n0, n0, __builtin_elementwise_fma(n1, n1, f32x2{1.0f, 1.0f}));
        const f32x2 dot = __builtin_elementwise_fma(n0, f32x2{f.s0, f.s0},
                                                    __builtin_elementwise_fma(n1, f32x2{f.s1, f.s1}, f32x2{f.s2, f.s2}));
        const f32x2 v = __builtin_elementwise_fma(n0, f32x2{f.sin_az, f.sin_az},
                                                  __builtin_elementwise_fma(n1, f32x2{f.cos_az, f.cos_az}, f32x2{-f.t_c, -f.t_c}));
        const f32x2 r = {__builtin_amdgcn_rsqf(S.x), __builtin_amdgcn_rsqf(S.y)};
        const f32x2 u = __builtin_elementwise_fma(dot, r, f32x2{-f.q_c, -f.q_c});
        f32x2 e = __builtin_elementwise_fma(S * r, f32x2{f.e_rel, f.e_rel}, f32x2{f.e_abs, f.e_abs});
        if (TINY) {       // exact zero differences give an exact t = 0; anything else gets an absolute term
#pragma unroll
            for (int h = 0; h < 2; ++h)
                e[h] = f.et_rel * __builtin_fmaf(__builtin_fabsf(n0[h]), __builtin_fabsf(f.sin_az),
                                                 __builtin_fabsf(n1[h]) * __builtin_fabsf(f.cos_az)) +
                       __builtin_fminf((__builtin_fabsf(dx[p][h]) + __builtin_fabsf(dy[p][h])) * 0x1p100f, 0x1p-120f);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool sane = S[h] < 0x1p60f;                                     // false for NaN / inf / huge
            const float au = __builtin_fabsf(u[h]), av = __builtin_fabsf(v[h]);
            const bool inc_yes = au <= f.q_h_in, inc_no = au > f.q_h_out;
            const bool back_known = av > e[h], back_no = back_known & (v[h] > 0.0f);
            // t_tiny: |v| > e never holds for an exact zero (e = 0, v = 0): decide that case by v <= 0 directly
            const bool back_yes = TINY ? ((back_known & (v[h] < 0.0f)) | ((e[h] == 0.0f) & (v[h] <= 0.0f)))
                                           : (back_known & !(v[h] > 0.0f));
            // result = low_inc | !backslope: known 1 if either says so, known 0 if both deny
            const bool one = sane & (inc_yes | back_no);
            const bool zero = sane & inc_no & back_yes;
            const int k = 2 * p + h;
            out |= (one ? 1u : 0u) << (8 * k);
            unsure |= ((one | zero) ? 0u : 1u) << k;
        }
    }
    if (unsure) {                       // rare: within the error bound of a threshold, or not finite
        const float cx[6] = {c0.y, c1.x, c1.y, c2.x, c2.y, c3.x};
        const float up[4] = {u0.x, u0.y, u1.x, u1.y}, dn[4] = {b0.x, b0.y, b1.x, b1.y};
        // one copy of the ~100-instruction exact path per quad, not four: it is cold code
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            if (!((unsure >> k) & 1u)) continue;
            const bool v = shadow_px_exact<F32>(a, cx[k + 2] - cx[k], 0.5f, dn[k] - up[k], 0.5f);
            out = (out & ~(0xffu << (8 * k))) | ((v ? 1u : 0u) << (8 * k));
        }
    }
    return out;
}

// One wave = 64 quads x SHADOW_ROWS consecutive output rows: every DEM row is loaded once per wave (eight
// floats per lane, 8-byte aligned) and serves as the row below, the centre row and the row above of successive
// output rows.  Measured on MI355X (ms per 3660^2 tile, +-4 % between boxes): SHADOW_ROWS 1: 0.0149, 2: 0.0125-0.0131,
// 3: 0.0127-0.0131, 4: 0.0133-0.0138, 6: 0.0135, 8: 0.0139, 16: 0.0157 -- short waves with all their loads in
// flight beat long ones, although a block then re-reads more halo rows (vertically adjacent blocks run on
// different XCDs, so the halo comes from the Infinity Cache / HBM, not from an L2).  An XCD-contiguous block order
// (guide T1: each XCD walks a run of row-blocks down a column strip, so halos meet in one L2) was measured too:
// 0.0141 -> 0.0157 ms per tile (0.0149 with the remap per tile instead of per launch) -- slower, as for the
// fused kernel in round 1; the plain order stays.  Waves per block (stacked in y; ms per tile, legacy promotion):
// 1: 0.0177, 2: 0.0133, 4: 0.0126, 8: 0.0133, 16: 0.0147 -- taller blocks save halo re-reads but run slower.
// Halving the L1 traffic does not help either: a variant in which every lane loads only its own 16 bytes of a row
// and takes the two halo values of the centre row from its neighbours through DPP wave shifts (one 16-byte + one
// 4-byte load per row instead of four 8-byte ones) measured 0.0140-0.0143 against 0.0131-0.0135 in the same process.
// What the kernel runs into is issue: ~196 VALU (two v_rsq_f32 per pixel pair at quarter rate among them) and
// ~130 SALU instructions per wave of 512 pixels keep the SIMDs' VALU 63 % busy (profiles/r02_next_rows_pmc.json:
// SQ_ACTIVE_INST_VALU x 4 / SIMD cycles) while the waves are short -- memory and arithmetic no longer overlap fully.
// Nor does trading SALU for VALU: the filter's decisions in sign-bit form (three packed differences per pixel pair,
// v_bitop3 / v_perm per pixel instead of v_cmp -> SGPR pair -> s_and / s_or -> v_cndmask chains: same VALU count,
// 25 SALU fewer per row) measured 0.0138-0.0143 against 0.0130-0.0133 for this compare form in one process.
// Round 3, the last structure not yet tried -- long waves WITH deep prefetch (a wave marches down 16 ... 128 output rows
// of its 256-pixel column strip with 3 ... 5 DEM rows in flight ahead of the row it computes, the row window rotating by
// register renaming; halo re-reads fall to 2 rows per strip segment): bit-identical, and 0.0145 (16 rows, 3 ahead) /
// 0.0153 (32, 3) / 0.0156 - 0.0170 (32, 5) / 0.0166 - 0.0176 (64, 5) / 0.0172 - 0.0184 (128, 5) against 0.0130 - 0.0131 for
// this kernel in the same process.  Longer is slower and deeper is slower: the registers of the rows in flight cost
// resident waves, and the halo traffic they save was never what the kernel waited for.
// (The waves of a block MUST be stacked in y: they share their halo rows through the CU's L1.  Numbering the
// work items along the rows instead -- no idle lanes at the row ends -- measured 0.0173.)
constexpr int SHADOW_ROWS = 2, SHADOW_WAVES = 4;     // waves (stacked in y) per block

// TINY: the slope threshold is (almost) zero -- see ShadowFilter::t_tiny
template <bool F32, bool TINY>
__global__ __launch_bounds__(64 * SHADOW_WAVES) void dswx_shadow_v3(const ShadowArgs a, const ShadowFilter f) {
    const int W = (int)a.width, H = (int)a.height, margin = (int)a.margin;
    const int ow = W - 2 * margin, oh = H - 2 * margin;
    const int oq = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy0 = ((a.by_first + (int)blockIdx.y * a.by_step) * SHADOW_WAVES + (threadIdx.x >> 6)) * SHADOW_ROWS;
    if (4 * oq >= ow || oy0 >= oh) return;
    const int ox = 4 * oq + 4 <= ow ? 4 * oq : ow - 4;          // the last quad of a ragged row overlaps the one before it
    const float* __restrict__ col = a.dem + (size_t)blockIdx.z * (size_t)H * (size_t)W + (size_t)(ox + margin);
    uint8_t* __restrict__ dst = a.shadow + (size_t)blockIdx.z * (size_t)a.out_stride + (size_t)ox;
    struct Row { f32x2 v[4]; };         // d[x-2 .. x+5] (inside the row: margin >= 2)
    auto load_row = [&](int y) {
        const float* r = col + (size_t)(y < H ? y : H - 1) * (size_t)W;        // rows past the last output row: clamped, unused
        // plain (cacheable) loads: neighbouring lanes' 32-byte pieces overlap by half, the second touch must hit
        // the cache (non-temporal loads measured 8 % slower)
        return Row{{*reinterpret_cast<const f32x2_u*>(r - 2), *reinterpret_cast<const f32x2_u*>(r),
                    *reinterpret_cast<const f32x2_u*>(r + 2), *reinterpret_cast<const f32x2_u*>(r + 4)}};
    };
    const int y0 = oy0 + margin;
    // software pipeline: the rows of output row i + 2 are requested before output row i is computed, so a
    // wave always has two DEM rows in flight (left to the compiler, every row's load sat directly in front of
    // its first use -- behind the branch of the exact path -- and the wave stalled a full latency per row)
    Row up = load_row(y0 - 1), ce = load_row(y0), dn = load_row(y0 + 1), nx = SHADOW_ROWS > 1 ? load_row(y0 + 2) : dn;
#pragma unroll
    for (int i = 0; i < SHADOW_ROWS; ++i) {
        const Row nn = i + 3 <= SHADOW_ROWS ? load_row(y0 + i + 3) : nx;       // only rows an output row of this wave needs
        if (oy0 + i < oh) {
            const uint32_t out = shadow_quad<F32, TINY>(a, f, ce.v[0], ce.v[1], ce.v[2], ce.v[3], up.v[1], up.v[2], dn.v[1], dn.v[2]);
            __builtin_nontemporal_store(out, reinterpret_cast<u32_u*>(dst + (size_t)(oy0 + i) * (size_t)ow));
        }
        up = ce;
        ce = dn;
        dn = nx;
        nx = nn;
    }
}

// ------------------------------------------------------------------------------
// LAND layer (row f3): the per-pixel part of create_landcover_mask :994-1115.
// One thread per HLS pixel: 3x3 WorldCover block -> three counts -> class hierarchy.
// ------------------------------------------------------------------------------
struct LandArgs {
    const uint8_t* wc3;      // [3H][3W]
    const uint8_t* cgls;     // [H][W]
    uint8_t* land;           // [H][W]
    long long height, width;
    long long out_stride;    // bytes between the LAND rasters of consecutive tiles (H W unless a batch says otherwise)
    uint32_t forest_bits[8]; // 256-bit set of CGLS forest classes
    int thr_tree, thr_low, thr_high, thr_water;
    int low_class, high_class;   // year_offset, 100 + year_offset (as uint8)
};

// `is_forest`: the CGLS class of the pixel is one of forest_mask_landcover_classes
__device__ __forceinline__ int land_class(const LandArgs& a, int water, int urban, int tree, bool is_forest) {
    if (!is_forest) tree = 0;
    int v = 255;
    if (tree >= a.thr_tree) v = 201;
    if (urban >= a.thr_low) v = a.low_class;
    if (urban >= a.thr_high) v = a.high_class;
    if (water >= a.thr_water) v = 200;
    return v;
}

// The nine class tests per pixel (is the byte 80 / 90 / 95, 50, 10?) are ONE 256-entry LDS table
// lookup per WorldCover byte -- water | urban << 4 | tree << 8, so that the sum of nine entries IS the
// three 3x3 counts (each <= 9 fits its 4-bit field).  Round 1 compared every byte against the five
// codes in registers: 116 VALU per pixel, issue-bound at 0.43 of the HBM rate; this form needs ~40.
// blockIdx.z = tile.  HBM-bound: 10 B read + 1 B written per pixel.
constexpr int LAND_ROWS = 1;

// Four HLS pixels x LAND_ROWS rows per thread: three 12-byte row pieces of the WorldCover map per row (a wave
// reads 768 contiguous bytes per row), one dword of CGLS, one dword stored.  width % 4 == 0.
// (Eight pixels per thread -- 1536-byte pieces, 72 VGPRs -- measured 9 % slower.)
__global__ __launch_bounds__(256) void dswx_landcover_v3(const LandArgs a) {
    __shared__ uint16_t s_code[256];
    __shared__ uint8_t s_forest[256];     // CGLS class -> is a forest class (a dynamic index into the kernel
                                          // argument's bit set is a global load in the middle of the pixel loop)
    {
        const int v = threadIdx.x;
        s_code[v] = (uint16_t)((((v == 80) | (v == 90) | (v == 95)) ? 1 : 0) | (v == 50 ? 16 : 0) | (v == 10 ? 256 : 0));
        s_forest[v] = (uint8_t)((a.forest_bits[v >> 5] >> (v & 31)) & 1u);
    }
    __syncthreads();
    // quads are numbered along the rows of the whole raster (LAND_ROWS = 1): consecutive waves continue along the
    // row, so the cache lines two 768-byte pieces share are fetched by one CU, and no lane is idle at the row ends
    const unsigned quads_per_row = (unsigned)(a.width >> 2);
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    const unsigned yq = q / quads_per_row;
    const long long xq = q - yq * quads_per_row, y0 = yq;
    if (y0 >= a.height) return;
    const long long W3 = 3 * a.width, tile = blockIdx.z;
    const uint8_t* wc = a.wc3 + tile * 9 * a.height * a.width;
    // all loads issued before the first lookup (more bytes in flight per wave); cacheable: neighbouring waves
    // share the cache lines at the ends of their pieces (non-temporal loads fetched them twice: 10 % slower)
    uint32_t w[LAND_ROWS][3][3], cg[LAND_ROWS];
#pragma unroll
    for (int r = 0; r < LAND_ROWS; ++r) {
        const long long y = y0 + r < a.height ? y0 + r : a.height - 1;          // past the last row: clamped, unused
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t* row = reinterpret_cast<const uint32_t*>(wc + (3 * y + i) * W3 + 12 * xq);
#pragma unroll
            for (int k = 0; k < 3; ++k) w[r][i][k] = row[k];
        }
        cg[r] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.cgls + tile * a.height * a.width + y * a.width + 4 * xq));
    }
#pragma unroll
    for (int r = 0; r < LAND_ROWS; ++r) {
        if (y0 + r >= a.height) break;
        uint32_t cnt[4] = {0u, 0u, 0u, 0u};      // per pixel: water | urban << 4 | tree << 8
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) cnt[j / 3] += s_code[(w[r][i][j >> 2] >> (8 * (j & 3))) & 0xffu];
        }
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            out |= (uint32_t)land_class(a, cnt[k] & 15u, (cnt[k] >> 4) & 15u, cnt[k] >> 8,
                                        s_forest[(cg[r] >> (8 * k)) & 0xffu] != 0) << (8 * k);
        __builtin_nontemporal_store(out, reinterpret_cast<uint32_t*>(a.land + tile * a.out_stride + (y0 + r) * a.width + 4 * xq));
    }
}

__global__ __launch_bounds__(256) void dswx_landcover_v1(const LandArgs a) {
    const long long x = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    const long long y = (long long)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.width || y >= a.height) return;
    int water = 0, urban = 0, tree = 0;
    const long long W3 = 3 * a.width;
    const long long tile = blockIdx.z;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint8_t* row = a.wc3 + tile * 9 * a.height * a.width + (3 * y + i) * W3 + 3 * x;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int v = row[j];
            water += (v == 80) | (v == 90) | (v == 95);
            urban += v == 50;
            tree += v == 10;
        }
    }
    const long long o = tile * a.height * a.width + y * a.width + x;
    const int c = a.cgls[o];
    a.land[tile * a.out_stride + y * a.width + x] = (uint8_t)land_class(a, water, urban, tree, ((a.forest_bits[c >> 5] >> (c & 31)) & 1u) != 0);
}

// ------------------------------------------------------------------------------
// Synthetic tiles (same integer recipe as proteus_amd/synth.py)
// ------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ long long fieldu(unsigned long long h, int shift, int bits) {
    return (long long)((h >> shift) & ((1ull << bits) - 1ull));
}

__global__ __launch_bounds__(256) void dswx_synth_v1(dswx_planes_in_t in, unsigned long long seed,
                                                      long long tile0, long long n_pixels, int width,
                                                      long long tile_stride) {
    const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
    if (px >= n_pixels) return;
    const long long t = blockIdx.y;
    const unsigned long long tile = (unsigned long long)(tile0 + t);
    const long long off = t * tile_stride + px;
    const unsigned long long K0 = 0x9E3779B97F4A7C15ull, K1 = 0xD1B54A32D192ED03ull;
    const unsigned long long h0 = mix64(seed * K0 + tile * K1 + (unsigned long long)px);
    const unsigned long long h1 = mix64(h0 + K0);
    const unsigned long long h2 = mix64(h1 + K0);
    const int cuts[5] = {14418, 26214, 42598, 55705, 64225};
    const int mean[5][6] = {{350, 450, 350, 250, 150, 100},
                            {500, 700, 600, 1300, 800, 500},
                            {300, 600, 400, 3500, 1800, 900},
                            {900, 1200, 1500, 2200, 2800, 2300},
                            {6000, 6200, 6400, 6600, 3000, 2500}};
    const int amps[5] = {300, 600, 600, 800, 2500};
    const int draw = (int)fieldu(h0, 0, 16);
    int stype = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) stype += draw >= cuts[k] ? 1 : 0;
    const bool is_fill = stype == 5;
    const int st = stype < 4 ? stype : 4;
    const long long amp = amps[st];
    const bool clip_evt = fieldu(h0, 16, 7) == 0;
    const int clip_band = (int)((fieldu(h0, 23, 3) * 6) >> 3);
    const long long clip_val = -fieldu(h0, 26, 8);
    const bool wrap_evt = fieldu(h0, 34, 10) == 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const long long noise = fieldu(h1, 10 * b, 10);
        long long v = mean[st][b] + ((noise * 2 * amp) >> 10) - amp;
        if (wrap_evt && (b == 1 || b == 4)) v += 19000;
        if (clip_evt && clip_band == b) v = clip_val;
        if (is_fill) v = -9999;
        const_cast<int16_t*>(in.band[b])[off] = (int16_t)v;
    }
    const long long aerosol = fieldu(h2, 0, 2);
    const long long water = fieldu(h2, 2, 5) < 10, snow = fieldu(h2, 7, 5) < 2,
                    shadow = fieldu(h2, 12, 5) < 3, adjacent = fieldu(h2, 17, 5) < 3,
                    cloud = fieldu(h2, 22, 5) < 4, cirrus = fieldu(h2, 27, 5) < 1;
    long long fm = (aerosol << 6) | (water << 5) | (snow << 4) | (shadow << 3) | (adjacent << 2) |
                   (cloud << 1) | cirrus;
    if (is_fill) fm = 255;
    const_cast<uint8_t*>(in.fmask)[off] = (uint8_t)fm;
    if (in.land) {
        const int classes[8] = {200, 201, 21, 121, 50, 150, 99, 100};
        const int cls = classes[fieldu(h2, 40, 3)];
        const_cast<uint8_t*>(in.land)[off] = (uint8_t)(fieldu(h2, 32, 8) < 179 ? 255 : cls);
    }
    if (in.shad) const_cast<uint8_t*>(in.shad)[off] = (uint8_t)(fieldu(h2, 43, 5) >= 3 ? 1 : 0);
    if (in.ocean) {
        const unsigned long long row_band = (unsigned long long)(px / width) >> 5;
        const unsigned long long hb = mix64(seed * K1 + tile * K0 + row_band + 0x5851F42D4C957F2Dull);
        const_cast<uint8_t*>(in.ocean)[off] = (uint8_t)(fieldu(hb, 0, 8) >= 13 ? 1 : 0);
    }
}

// ==============================================================================
// host side
// ==============================================================================
extern "C" {

int dswx_interpret_layer_host(dswx_ctx_t* ctx, const int64_t* diag_decimal, int64_t n, uint8_t* out) {
    if (!ctx || (n > 0 && (!diag_decimal || !out))) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (n < 0) return dswx_fail(DSWX_ERR_ARG, "negative size");
    if (n == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    void* d_in = nullptr;
    void* d_out = nullptr;
    HIP_TRY(dswx_locked_malloc(&d_in, (size_t)n * 8));
    hipError_t e = dswx_locked_malloc(&d_out, (size_t)n);
    if (e != hipSuccess) { (void)hipFree(d_in); return dswx_fail(DSWX_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e)); }
    hipStream_t s = ctx->stream;
    e = hipMemcpyAsync(d_in, diag_decimal, (size_t)n * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(dswx_interpret_v1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                           static_cast<const long long*>(d_in), static_cast<uint8_t*>(d_out), (long long)n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return dswx_fail(DSWX_ERR_HIP, "dswx_interpret_layer_host: %s", hipGetErrorString(e));
    return DSWX_OK;
}

// ---- terrain shadow: thresholds on the arccos / arctan ARGUMENTS --------------------------
// doubles in numeric order as int64 (both zeros map to 0)
static int64_t ord_of(double d) {
    int64_t i;
    std::memcpy(&i, &d, 8);
    return i >= 0 ? i : -(i & 0x7fffffffffffffffLL);
}
static double dbl_of(int64_t k) {
    int64_t i = k >= 0 ? k : (int64_t)(0x8000000000000000ULL | (uint64_t)(-k));
    double d;
    std::memcpy(&d, &i, 8);
    return d;
}
static const double kRad2Deg = 180.0 / 3.141592653589793238462643383279502884;

int dswx_shadow_thresholds(double min_slope_angle, double max_sun_local_inc_angle, double* slope_arg_max,
                           double* inc_q_min) {
    if (!slope_arg_max || !inc_q_min) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (std::isnan(min_slope_angle) || std::isnan(max_sun_local_inc_angle))
        return dswx_fail(DSWX_ERR_ARG, "shadow angle threshold is NaN");
    // smallest q in [-1, 1] with degrees(acos(q)) <= T (acos decreases); 2.0 if there is none
    auto inc_ok = [&](double q) { return std::acos(q) * kRad2Deg <= max_sun_local_inc_angle; };
    if (!inc_ok(1.0)) *inc_q_min = 2.0;
    else if (inc_ok(-1.0)) *inc_q_min = -1.0;
    else {
        int64_t lo = ord_of(-1.0), hi = ord_of(1.0);          // !ok(lo), ok(hi)
        while ((uint64_t)hi - (uint64_t)lo > 1) {
            const int64_t mid = lo + (int64_t)(((uint64_t)hi - (uint64_t)lo) / 2);
            if (inc_ok(dbl_of(mid))) hi = mid; else lo = mid;
        }
        *inc_q_min = dbl_of(hi);
    }
    // largest t with degrees(atan(t)) <= T (atan increases); +-inf when always / never
    auto slope_ok = [&](double t) { return std::atan(t) * kRad2Deg <= min_slope_angle; };
    const double inf = std::numeric_limits<double>::infinity();
    if (slope_ok(inf)) *slope_arg_max = inf;
    else if (!slope_ok(-inf)) *slope_arg_max = -inf;
    else {
        int64_t lo = ord_of(-inf), hi = ord_of(inf);          // ok(lo), !ok(hi); the span exceeds int64
        while ((uint64_t)hi - (uint64_t)lo > 1) {
            const int64_t mid = lo + (int64_t)(((uint64_t)hi - (uint64_t)lo) / 2);
            if (slope_ok(dbl_of(mid))) lo = mid; else hi = mid;
        }
        *slope_arg_max = dbl_of(lo);
    }
    return DSWX_OK;
}

static int shadow_args(ShadowArgs* a, int64_t height, int64_t width, int64_t margin, const double sun_vector[3],
                       double sin_azimuth, double cos_azimuth, double slope_arg_max, double inc_q_min,
                       double pixel_spacing_x, double pixel_spacing_y) {
    if (!sun_vector) return dswx_fail(DSWX_ERR_ARG, "sun_vector is NULL");
    if (height < 2 || width < 2)
        return dswx_fail(DSWX_ERR_ARG, "Shape of array too small to calculate a numerical gradient, "
                                  "at least 2 elements are required.");
    if (margin < 0 || 2 * margin >= height || 2 * margin >= width) return dswx_fail(DSWX_ERR_ARG, "bad margin");
    if (height > 2147483647LL / width) return dswx_fail(DSWX_ERR_ARG, "DEM larger than 2^31 pixels");
    if (std::isnan(slope_arg_max) || std::isnan(inc_q_min)) return dswx_fail(DSWX_ERR_ARG, "shadow threshold is NaN");
    a->height = height; a->width = width; a->margin = margin;
    a->spacing_x = (float)pixel_spacing_x;
    a->neg_abs_spacing_y = (float)(-std::fabs(pixel_spacing_y));
    for (int i = 0; i < 3; ++i) a->sun[i] = sun_vector[i];
    a->sin_az = sin_azimuth; a->cos_az = cos_azimuth;
    a->slope_arg_max = slope_arg_max; a->inc_q_min = inc_q_min;
    a->by_first = 0; a->by_step = 1;
    return DSWX_OK;
}

// largest float <= v / smallest float >= v (v may be +-inf; NaN never reaches here)
static float float_down(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
static float float_up(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

// Constants of the approximate evaluation in front of the exact one (dswx_shadow_v3).  In float32
// mode the thresholds the exact path compares against are the float32 roundings of a.inc_q_min /
// a.slope_arg_max (already float32 values when they come through the _q32 entry points).
static void shadow_filter(const ShadowArgs& a, bool f32, ShadowFilter* f) {
    const double q_min = f32 ? (double)(float)a.inc_q_min : a.inc_q_min;
    const double t_max = f32 ? (double)(float)a.slope_arg_max : a.slope_arg_max;
    f->inv_x = (float)(1.0 / (-2.0 * (double)a.spacing_x));
    f->inv_y = (float)(1.0 / (-2.0 * (double)a.neg_abs_spacing_y));
    f->s0 = (float)a.sun[0]; f->s1 = (float)a.sun[1]; f->s2 = (float)a.sun[2];
    f->sin_az = (float)a.sin_az; f->cos_az = (float)a.cos_az;
    // low_inc <=> q_min <= q <= 1, tested as |q - c| against half the interval: |q~ - q_ref| <= K (see the
    // kernel comment), + the float32 roundings of the centre and of the subtraction
    const double sum_abs = std::fabs(a.sun[0]) + std::fabs(a.sun[1]) + std::fabs(a.sun[2]);
    const double K = std::ldexp(sum_abs, -18) + std::ldexp(sum_abs + 2.0, -22) + 1e-30;
    f->q_c = (float)(0.5 * (q_min + 1.0));
    const double hw_lo = (double)f->q_c - q_min, hw_hi = 1.0 - (double)f->q_c;
    f->q_h_in = float_down((hw_lo < hw_hi ? hw_lo : hw_hi) - K);
    f->q_h_out = float_up((hw_lo > hw_hi ? hw_lo : hw_hi) + K);
    // backslope <=> t <= t_max, tested as v = t~ - t_c against e >= |t~ - t_ref| + |t_c - t_max|:
    // 2^-19 (|n0 sin| + |n1 cos|) covers the arithmetic (7 half-ulps of float32 needed, 4 x slack, which
    // also pays for the rounding of the subtraction), and |n0 sin| + |n1 cos| <= sqrt(S) |(sin, cos)|
    const double hyp = std::sqrt(a.sin_az * a.sin_az + a.cos_az * a.cos_az);
    f->e_rel = float_up(std::ldexp(hyp, -19));
    f->et_rel = 0x1p-19f;
    f->t_tiny = std::fabs(t_max) < 0x1p-100 ? 1 : 0;
    if (std::isinf(t_max) || std::fabs(t_max) > 3e38) {
        f->t_c = t_max > 0 ? std::numeric_limits<float>::infinity() : -std::numeric_limits<float>::infinity();
        f->e_abs = 0.0f;                 // v = -+inf (or NaN): decided by its sign, NaN falls to the exact path
    } else {
        f->t_c = (float)t_max;
        // float32 rounding of the threshold + of the subtraction's threshold share + denormal noise (< 2^-140)
        f->e_abs = f->t_tiny ? 0.0f : float_up(std::fabs((double)f->t_c - t_max) + std::fabs(t_max) * 0x1p-20 + 1e-40);
    }
}

static int shadow_device_impl(dswx_ctx_t* ctx, const float* dem, int64_t n_tiles, int64_t height, int64_t width,
                              int64_t margin, const double sun_vector[3], double sin_azimuth,
                              double cos_azimuth, double slope_arg_max, double inc_q_min,
                              double pixel_spacing_x, double pixel_spacing_y, uint8_t* shadow, void* stream,
                              bool f32, int64_t shadow_tile_stride = 0) {
    if (!ctx || !dem || !shadow) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (n_tiles < 0 || n_tiles > 65535) return dswx_fail(DSWX_ERR_ARG, "n_tiles out of range");
    ShadowArgs a;
    int rc = shadow_args(&a, height, width, margin, sun_vector, sin_azimuth, cos_azimuth, slope_arg_max, inc_q_min,
                         pixel_spacing_x, pixel_spacing_y);
    if (rc) return rc;
    if (n_tiles == 0) return DSWX_OK;
    a.dem = dem; a.shadow = shadow;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const long long ow = width - 2 * margin, oh = height - 2 * margin;
    if (shadow_tile_stride != 0 && shadow_tile_stride < oh * ow)
        return dswx_fail(DSWX_ERR_ARG, "shadow_tile_stride smaller than the shadow raster");
    a.out_stride = shadow_tile_stride ? shadow_tile_stride : oh * ow;
    if ((oh + 3) / 4 > 65535) return dswx_fail(DSWX_ERR_ARG, "raster too tall for one launch");
    // four pixels per thread behind the filter wherever a quad and the columns beside it are interior pixels (unaligned
    // 8-byte loads / dword stores: any margin >= 2, any width, any buffer alignment -- see dswx_shadow_v3)
    const bool quads = margin >= 2 && ow >= 4 && aligned_to(dem, 4) && ctx->shadow_kernel != 2;
    if (quads) {
        ShadowFilter f;
        shadow_filter(a, f32, &f);
        // Lab A/B (shadow_grid_pad): grid.x rounded up to a multiple of 8, surplus blocks returning at once.
        // Workgroups are dealt to the eight XCDs round-robin by linear block id, so with 8 | grid.x the block below
        // (id + grid.x) runs on the SAME XCD a moment later and finds the two halo rows it shares with this one
        // in that XCD's L2.  Measured: 0.0132 (as it comes, grid.x = 15) / 0.0133 (16) ms per tile -- nothing: the
        // 1.27 x over-fetch the L2 counters show is absorbed behind the L2 (Infinity Cache), it is not what the
        // kernel waits for.  Default 1 = as it comes.
        const long long pad = ctx->shadow_grid_pad > 0 ? ctx->shadow_grid_pad : 1;
        const long long gx = (((ow + 3) / 4 + 63) / 64 + pad - 1) / pad * pad;
        const unsigned block_rows = (unsigned)((oh + SHADOW_WAVES * SHADOW_ROWS - 1) / (SHADOW_WAVES * SHADOW_ROWS));
        const dim3 block(64 * SHADOW_WAVES);
        const int passes = ctx->shadow_kernel == 3 ? 2 : 1;      // lab: even block rows of all tiles, then the odd ones
        for (int pass = 0; pass < passes; ++pass) {
            a.by_first = pass; a.by_step = passes;
            const unsigned gy = (block_rows - (unsigned)pass + (unsigned)passes - 1) / (unsigned)passes;
            if (gy == 0) continue;
            const dim3 grid((unsigned)gx, gy, (unsigned)n_tiles);
            if (f32 && f.t_tiny) hipLaunchKernelGGL((dswx_shadow_v3<true, true>), grid, block, 0, s, a, f);
            else if (f32) hipLaunchKernelGGL((dswx_shadow_v3<true, false>), grid, block, 0, s, a, f);
            else if (f.t_tiny) hipLaunchKernelGGL((dswx_shadow_v3<false, true>), grid, block, 0, s, a, f);
            else hipLaunchKernelGGL((dswx_shadow_v3<false, false>), grid, block, 0, s, a, f);
        }
    } else {
        dim3 grid((unsigned)((ow + 63) / 64), (unsigned)((oh + 3) / 4), (unsigned)n_tiles), block(256);
        if (f32) hipLaunchKernelGGL(dswx_shadow_v2<true>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(dswx_shadow_v2<false>, grid, block, 0, s, a);
    }
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

int dswx_shadow_layer_device_q(dswx_ctx_t* ctx, const float* dem, int64_t n_tiles, int64_t height, int64_t width,
                               int64_t margin, const double sun_vector[3], double sin_azimuth,
                               double cos_azimuth, double slope_arg_max, double inc_q_min,
                               double pixel_spacing_x, double pixel_spacing_y, uint8_t* shadow, void* stream) {
    return shadow_device_impl(ctx, dem, n_tiles, height, width, margin, sun_vector, sin_azimuth, cos_azimuth,
                              slope_arg_max, inc_q_min, pixel_spacing_x, pixel_spacing_y, shadow, stream, false);
}

int dswx_shadow_layer_device_q32(dswx_ctx_t* ctx, const float* dem, int64_t n_tiles, int64_t height, int64_t width,
                                 int64_t margin, const double sun_vector[3], double sin_azimuth,
                                 double cos_azimuth, float slope_arg_max, float inc_q_min,
                                 double pixel_spacing_x, double pixel_spacing_y, uint8_t* shadow, void* stream) {
    return shadow_device_impl(ctx, dem, n_tiles, height, width, margin, sun_vector, sin_azimuth, cos_azimuth,
                              (double)slope_arg_max, (double)inc_q_min, pixel_spacing_x, pixel_spacing_y, shadow,
                              stream, true);
}

int dswx_shadow_layer_device(dswx_ctx_t* ctx, const float* dem, int64_t n_tiles, int64_t height, int64_t width,
                             int64_t margin, const double sun_vector[3], double sin_azimuth,
                             double cos_azimuth, double min_slope_angle, double max_sun_local_inc_angle,
                             double pixel_spacing_x, double pixel_spacing_y, uint8_t* shadow, void* stream) {
    double slope_arg_max, inc_q_min;
    int rc = dswx_shadow_thresholds(min_slope_angle, max_sun_local_inc_angle, &slope_arg_max, &inc_q_min);
    if (rc) return rc;
    return dswx_shadow_layer_device_q(ctx, dem, n_tiles, height, width, margin, sun_vector, sin_azimuth, cos_azimuth,
                                      slope_arg_max, inc_q_min, pixel_spacing_x, pixel_spacing_y, shadow, stream);
}

static int shadow_host_impl(dswx_ctx_t* ctx, const float* dem, int64_t height, int64_t width, int64_t margin,
                            const double sun_vector[3], double sin_azimuth, double cos_azimuth,
                            double slope_arg_max, double inc_q_min, double pixel_spacing_x,
                            double pixel_spacing_y, uint8_t* shadow, bool f32) {
    if (!ctx || !dem || !shadow) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    ShadowArgs chk;
    int rc = shadow_args(&chk, height, width, margin, sun_vector, sin_azimuth, cos_azimuth, slope_arg_max, inc_q_min,
                         pixel_spacing_x, pixel_spacing_y);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t in_bytes = (size_t)height * (size_t)width * 4;
    const size_t out_px = (size_t)(height - 2 * margin) * (size_t)(width - 2 * margin);
    void* d_dem = nullptr;
    void* d_out = nullptr;
    HIP_TRY(dswx_locked_malloc(&d_dem, in_bytes));
    hipError_t e = dswx_locked_malloc(&d_out, out_px);
    hipStream_t s = ctx->stream;
    if (e == hipSuccess) e = hipMemcpyAsync(d_dem, dem, in_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        rc = shadow_device_impl(ctx, static_cast<const float*>(d_dem), 1, height, width, margin, sun_vector,
                                sin_azimuth, cos_azimuth, slope_arg_max, inc_q_min, pixel_spacing_x,
                                pixel_spacing_y, static_cast<uint8_t*>(d_out), s, f32);
        if (rc == DSWX_OK) e = hipMemcpyAsync(shadow, d_out, out_px, hipMemcpyDeviceToHost, s);
        if (rc == DSWX_OK && e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void)hipFree(d_dem);
    if (d_out) (void)hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) return dswx_fail(DSWX_ERR_HIP, "dswx_shadow_layer_host: %s", hipGetErrorString(e));
    return DSWX_OK;
}

int dswx_shadow_layer_host_q(dswx_ctx_t* ctx, const float* dem, int64_t height, int64_t width, int64_t margin,
                             const double sun_vector[3], double sin_azimuth, double cos_azimuth,
                             double slope_arg_max, double inc_q_min, double pixel_spacing_x,
                             double pixel_spacing_y, uint8_t* shadow) {
    return shadow_host_impl(ctx, dem, height, width, margin, sun_vector, sin_azimuth, cos_azimuth, slope_arg_max,
                            inc_q_min, pixel_spacing_x, pixel_spacing_y, shadow, false);
}

int dswx_shadow_layer_host_q32(dswx_ctx_t* ctx, const float* dem, int64_t height, int64_t width, int64_t margin,
                               const double sun_vector[3], double sin_azimuth, double cos_azimuth,
                               float slope_arg_max, float inc_q_min, double pixel_spacing_x,
                               double pixel_spacing_y, uint8_t* shadow) {
    return shadow_host_impl(ctx, dem, height, width, margin, sun_vector, sin_azimuth, cos_azimuth,
                            (double)slope_arg_max, (double)inc_q_min, pixel_spacing_x, pixel_spacing_y, shadow, true);
}

int dswx_shadow_layer_host(dswx_ctx_t* ctx, const float* dem, int64_t height, int64_t width, int64_t margin,
                           const double sun_vector[3], double sin_azimuth, double cos_azimuth,
                           double min_slope_angle, double max_sun_local_inc_angle, double pixel_spacing_x,
                           double pixel_spacing_y, uint8_t* shadow) {
    double slope_arg_max, inc_q_min;
    int rc = dswx_shadow_thresholds(min_slope_angle, max_sun_local_inc_angle, &slope_arg_max, &inc_q_min);
    if (rc) return rc;
    return dswx_shadow_layer_host_q(ctx, dem, height, width, margin, sun_vector, sin_azimuth, cos_azimuth,
                                    slope_arg_max, inc_q_min, pixel_spacing_x, pixel_spacing_y, shadow);
}

static int land_args(LandArgs* a, int64_t height, int64_t width, const int32_t* forest_classes,
                     int32_t n_forest_classes, const int32_t thresholds[4], int32_t year_offset) {
    if (!thresholds) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (height < 0 || width < 0 || n_forest_classes < 0 || (n_forest_classes > 0 && !forest_classes))
        return dswx_fail(DSWX_ERR_ARG, "bad size");
    std::memset(a, 0, sizeof *a);
    for (int i = 0; i < n_forest_classes; ++i) {
        const int c = forest_classes[i];
        if (c >= 0 && c <= 255) a->forest_bits[c >> 5] |= 1u << (c & 31);
    }
    a->thr_tree = thresholds[0]; a->thr_low = thresholds[1]; a->thr_high = thresholds[2]; a->thr_water = thresholds[3];
    // numpy stores the class through a uint8 array: values wrap modulo 256
    a->low_class = (int)(uint8_t)(0 + year_offset);
    a->high_class = (int)(uint8_t)(100 + year_offset);
    a->height = height; a->width = width;
    return DSWX_OK;
}

static int landcover_device_impl(dswx_ctx_t* ctx, const uint8_t* worldcover_up3, const uint8_t* copernicus,
                                 int64_t n_tiles, int64_t height, int64_t width, const int32_t* forest_classes,
                                 int32_t n_forest_classes, const int32_t thresholds[4], int32_t year_offset,
                                 uint8_t* land, int64_t land_tile_stride, void* stream);

int dswx_landcover_mask_device(dswx_ctx_t* ctx, const uint8_t* worldcover_up3, const uint8_t* copernicus,
                               int64_t n_tiles, int64_t height, int64_t width, const int32_t* forest_classes,
                               int32_t n_forest_classes, const int32_t thresholds[4], int32_t year_offset,
                               uint8_t* land, void* stream) {
    return landcover_device_impl(ctx, worldcover_up3, copernicus, n_tiles, height, width, forest_classes, n_forest_classes,
                                 thresholds, year_offset, land, 0, stream);
}

int dswx_landcover_mask_batch(dswx_ctx_t* ctx, const uint8_t* worldcover_up3, const uint8_t* copernicus,
                              int64_t n_tiles, int64_t height, int64_t width, const int32_t* forest_classes,
                              int32_t n_forest_classes, const int32_t thresholds[4], int32_t year_offset,
                              uint8_t* land, int64_t land_tile_stride, void* stream) {
    return landcover_device_impl(ctx, worldcover_up3, copernicus, n_tiles, height, width, forest_classes, n_forest_classes,
                                 thresholds, year_offset, land, land_tile_stride, stream);
}

int dswx_shadow_layer_batch(dswx_ctx_t* ctx, const float* dem, int64_t n_tiles, int64_t height, int64_t width,
                            int64_t margin, const double sun_vector[3], double sin_azimuth, double cos_azimuth,
                            double slope_arg_max, double inc_q_min, int32_t float32_arithmetic, double pixel_spacing_x,
                            double pixel_spacing_y, uint8_t* shadow, int64_t shadow_tile_stride, void* stream) {
    return shadow_device_impl(ctx, dem, n_tiles, height, width, margin, sun_vector, sin_azimuth, cos_azimuth, slope_arg_max,
                              inc_q_min, pixel_spacing_x, pixel_spacing_y, shadow, stream, float32_arithmetic != 0,
                              shadow_tile_stride);
}

static int landcover_device_impl(dswx_ctx_t* ctx, const uint8_t* worldcover_up3, const uint8_t* copernicus,
                                 int64_t n_tiles, int64_t height, int64_t width, const int32_t* forest_classes,
                                 int32_t n_forest_classes, const int32_t thresholds[4], int32_t year_offset,
                                 uint8_t* land, int64_t land_tile_stride, void* stream) {
    if (!ctx || !worldcover_up3 || !copernicus || !land) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (land_tile_stride != 0 && land_tile_stride < height * width)
        return dswx_fail(DSWX_ERR_ARG, "land_tile_stride smaller than the raster");
    if (n_tiles < 0 || n_tiles > 65535) return dswx_fail(DSWX_ERR_ARG, "n_tiles out of range");
    LandArgs a;
    int rc = land_args(&a, height, width, forest_classes, n_forest_classes, thresholds, year_offset);
    if (rc) return rc;
    if (n_tiles == 0 || height == 0 || width == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    a.wc3 = worldcover_up3; a.cgls = copernicus; a.land = land;
    a.out_stride = land_tile_stride ? land_tile_stride : height * width;
    // four pixels per thread with dword loads when rows keep 4-byte alignment
    const bool quad = width % 4 == 0 && aligned_to(worldcover_up3, 4) && aligned_to(copernicus, 4) && aligned_to(land, 4) &&
                      a.out_stride % 4 == 0;
    if (quad) {
        static_assert(LAND_ROWS == 1, "the flat quad numbering assumes one row per thread");
        const long long quads = (width / 4) * height;
        if (quads > 0xffffffffLL - 256) return dswx_fail(DSWX_ERR_ARG, "raster too large for one launch");
        dim3 grid((unsigned)((quads + 255) / 256), 1, (unsigned)n_tiles), block(256);
        hipLaunchKernelGGL(dswx_landcover_v3, grid, block, 0, s, a);
    } else {
        dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4), (unsigned)n_tiles), block(256);
        if (grid.y > 65535) return dswx_fail(DSWX_ERR_ARG, "raster too tall for one launch");
        hipLaunchKernelGGL(dswx_landcover_v1, grid, block, 0, s, a);
    }
    HIP_TRY(hipGetLastError());
    return DSWX_OK;
}

int dswx_landcover_mask_host(dswx_ctx_t* ctx, const uint8_t* worldcover_up3, const uint8_t* copernicus,
                             int64_t height, int64_t width, const int32_t* forest_classes,
                             int32_t n_forest_classes, const int32_t thresholds[4], int32_t year_offset,
                             uint8_t* land) {
    if (!ctx || !worldcover_up3 || !copernicus || !thresholds || !land) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    {
        LandArgs chk;
        int rc = land_args(&chk, height, width, forest_classes, n_forest_classes, thresholds, year_offset);
        if (rc) return rc;
    }
    if (height == 0 || width == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)height * (size_t)width;
    void* d_wc = nullptr; void* d_cg = nullptr; void* d_out = nullptr;
    hipError_t e = dswx_locked_malloc(&d_wc, 9 * n);
    if (e == hipSuccess) e = dswx_locked_malloc(&d_cg, n);
    if (e == hipSuccess) e = dswx_locked_malloc(&d_out, n);
    hipStream_t s = ctx->stream;
    if (e == hipSuccess) e = hipMemcpyAsync(d_wc, worldcover_up3, 9 * n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cg, copernicus, n, hipMemcpyHostToDevice, s);
    int rc = DSWX_OK;
    if (e == hipSuccess)
        rc = dswx_landcover_mask_device(ctx, static_cast<const uint8_t*>(d_wc), static_cast<const uint8_t*>(d_cg), 1,
                                        height, width, forest_classes, n_forest_classes, thresholds, year_offset,
                                        static_cast<uint8_t*>(d_out), s);
    if (e == hipSuccess && rc == DSWX_OK) e = hipMemcpyAsync(land, d_out, n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (d_wc) (void)hipFree(d_wc);
    if (d_cg) (void)hipFree(d_cg);
    if (d_out) (void)hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) return dswx_fail(DSWX_ERR_HIP, "dswx_landcover_mask_host: %s", hipGetErrorString(e));
    return DSWX_OK;
}

static int synth_impl(dswx_ctx_t* ctx, uint64_t seed, int64_t tile0, int64_t n_tiles, int64_t height,
                      int64_t width, int64_t tile_stride, const dswx_planes_in_t* in, void* stream) {
    if (!ctx || !in) return dswx_fail(DSWX_ERR_ARG, "NULL argument");
    if (n_tiles < 0 || height < 0 || width < 0 || tile0 < 0) return dswx_fail(DSWX_ERR_ARG, "negative size");
    for (int k = 0; k < 6; ++k)
        if (!in->band[k]) return dswx_fail(DSWX_ERR_ARG, "band[%d] is NULL", k);
    if (!in->fmask) return dswx_fail(DSWX_ERR_ARG, "fmask is NULL");
    const int64_t P = height * width;
    if (tile_stride == 0) tile_stride = P;
    if (tile_stride < P) return dswx_fail(DSWX_ERR_ARG, "tile_stride smaller than the tile");
    if (n_tiles == 0 || P == 0) return DSWX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int64_t max_y = 65535;
    for (int64_t t0 = 0; t0 < n_tiles; t0 += max_y) {
        const int64_t nt = (n_tiles - t0 < max_y) ? n_tiles - t0 : max_y;
        dswx_planes_in_t b = *in;
        const int64_t shift = t0 * tile_stride;
        for (int k = 0; k < 6; ++k) b.band[k] += shift;
        b.fmask += shift;
        if (b.land) b.land += shift;
        if (b.shad) b.shad += shift;
        if (b.ocean) b.ocean += shift;
        dim3 grid((unsigned)((P + 255) / 256), (unsigned)nt), block(256);
        hipLaunchKernelGGL(dswx_synth_v1, grid, block, 0, s, b, (unsigned long long)seed,
                           (long long)(tile0 + t0), (long long)P, (int)width, (long long)tile_stride);
        HIP_TRY(hipGetLastError());
    }
    return DSWX_OK;
}

int dswx_synth_fill(dswx_ctx_t* ctx, uint64_t seed, int64_t tile0, int64_t n_tiles, int64_t height,
                    int64_t width, const dswx_planes_in_t* in, void* stream) {
    return synth_impl(ctx, seed, tile0, n_tiles, height, width, 0, in, stream);
}

int dswx_synth_batch(dswx_ctx_t* ctx, uint64_t seed, int64_t tile0, const dswx_batch_geom_t* geom,
                     const dswx_planes_in_t* in, void* stream) {
    if (!geom) return dswx_fail(DSWX_ERR_ARG, "geom is NULL");
    return synth_impl(ctx, seed, tile0, geom->n_tiles, geom->height, geom->width, geom->tile_stride, in, stream);
}

}  // extern "C"
