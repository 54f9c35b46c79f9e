// rtiow_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the
// per-pixel path-tracing hot path.
//
//   ch_kernel_rows          raytrace05.comp / raytrace06.comp (the reference's two compute shaders): four pixels of a row
//                           per lane, per-column / per-row terms shared through LDS, 16-byte stores
//   ch_kernel_tiles         the same, one lane per pixel in 16x16 tiles as the reference dispatches them (degenerate
//                           image sizes; RtParams.kernel = 1: the form the row kernel is checked against)
//   path_pixel_kernel       PATH mode v1 (RtParams.kernel 1): one lane per pixel, spp + bounce loops inside, sphere
//                           list staged in LDS once per workgroup; kept as a cross-check and ablation
//   path_persistent_kernel  PATH mode v2, eight instantiations <shading records in LDS?, clustered list?, flat-axis boxes?, compact per-wave area?>:
//                           persistent waves, two path slots per lane, per-XCD pixel queues, LDS accumulators, in-kernel
//                           resolve.  <., false, false> walks the flat sphere list (kernel 2; default below 64 spheres),
//                           <., true, .> the two-level clustered list (kernel 3; default from 64 spheres on) with camera
//                           rays traced in the primary pass (primary_trace: kernel 4 forces it at any spp); <., true, true>
//                           tests the cluster boxes without the axis they all share (scenes that stand on a plane); <false, true, ., true>
//                           is the large-scene variant at four waves per SIMD (launch_path takes it where it keeps more waves per CU)
//   order_chunks_kernel     the next frame's chunk sequence from this frame's per-chunk costs (cost-ordered dequeue)
//   arith_kernel            one operation per element: the arithmetic conformance probe of rtSelfTestArith
//   (deinterleave_kernel, the multi-GPU frame assembly, lives in rtiow_multi.hip)
//
// Arithmetic contract: binary32, round-to-nearest-even, denormals kept, only
// + - * / sqrt and fma; compiled with -ffp-contract=off so a fused multiply-add
// exists exactly where __builtin_fmaf is written (hipcc's default correctly
// rounded fp32 divide/sqrt stay on).  tests/ check every kernel bit-for-bit
// against the independent CPU restatement in oracle/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "rtiow_device.h"
#include "rtiow_rng.h"

// This file is compiled three times (Makefile): whole, and twice for one family of the clustered persistent kernels each, with a
// scheduler (and a register budget) of its own -- -DRTIOW_TU_SMALL_CLUSTERED: the small-scene variants, -DRTIOW_TU_LARGE_CLUSTERED:
// the large-scene ones.  Those two passes compile the shared device code and their kernels only (see the end of the file).
#if defined(RTIOW_TU_SMALL_CLUSTERED) || defined(RTIOW_TU_LARGE_CLUSTERED)
#define RTIOW_TU_PART
#endif

namespace rtiow {
namespace {

#define DI __device__ __forceinline__
#define HDI __host__ __device__ __forceinline__  // (also run on the host: rtConeSelfTestHost)

struct f3 {
    float x, y, z;
};

DI f3 mk(float x, float y, float z) { return f3{x, y, z}; }
HDI float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DI float dot3(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
DI f3 unit3(f3 a) {
    const float k = 1.0f / __builtin_sqrtf(dot3(a, a));
    return mk(a.x * k, a.y * k, a.z * k);
}
DI float psqrt(float x);
DI float precip(float s);
DI f3 unit3_scattered(f3 a) {  // the same for a direction the kernels made themselves (see psqrt)
    const float k = precip(psqrt(dot3(a, a)));
    return mk(a.x * k, a.y * k, a.z * k);
}

constexpr float kTMin = 0.001f;

// ---- quantisers + pack (a5, a10) -------------------------------------------
DI uint32_t quant_unorm8(float x) {
    const float c = (x > 0.0f) ? (x < 1.0f ? x : 1.0f) : 0.0f;
    return static_cast<uint32_t>(static_cast<int>(c * 255.0f + 0.5f));
}
DI uint32_t quant_book(float x) {
    const float c = (x > 0.0f) ? (x < 0.999f ? x : 0.999f) : 0.0f;
    return static_cast<uint32_t>(static_cast<int>(256.0f * c));
}
DI uint32_t pack_rgb(uint32_t r, uint32_t g, uint32_t b) { return r | (g << 8) | (b << 16); }

// Multi-sample accumulate (a10): each sample's radiance goes to unsigned 32.32 fixed
// point (clamp [0,32768], times 2^32 — exact — truncate) and pixels are INTEGER sums,
// so the result does not depend on which lane added which sample, or in what order.
// (Six instructions: trunc(c * 2^32) = floor(c) * 2^32 + floor(frac(c) * 2^32) for 0 <= c <= 32768 -- floor(c) and frac(c) = c - floor(c)
// are exact, the scaling by 2^32 is, and both parts fit 32 bits -- where the compiler's float -> u64 conversion of the same number is ten.
// fmax(NaN, 0) = 0: the clamp sends a NaN to 0 as the comparisons of the plain form do.)
DI unsigned long long to_fixed(float x) {
    const float c = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 32768.0f);  // NaN -> 0
    uint32_t hi, lo;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(hi) : "v"(c));  // (truncation; a C cast would be the same here, c being in range)
    const float f = __builtin_amdgcn_fractf(c) * 4294967296.0f;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(lo) : "v"(f));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

DI uint32_t resolve_pixel(unsigned long long sr, unsigned long long sg, unsigned long long sb,
                          uint32_t spp, uint32_t quantiser) {
    const float scale = 1.0f / (static_cast<float>(spp) * 4294967296.0f);
    const float r = psqrt(scale * static_cast<float>(sr));
    const float g = psqrt(scale * static_cast<float>(sg));
    const float b = psqrt(scale * static_cast<float>(sb));
    if (quantiser == RT_QUANT_BOOK) return pack_rgb(quant_book(r), quant_book(g), quant_book(b));
    return pack_rgb(quant_unorm8(r), quant_unorm8(g), quant_unorm8(b));
}

// Turns a pixel's sums of THIS dispatch into its colour.  Progressive accumulation: the context's
// accumulators hold the sums of the samples before a.sample_offset; the pixel is owned by exactly one
// lane per dispatch, so a plain read-modify-write suffices.
DI uint32_t close_pixel(const PathArgs& a, uint32_t local_pix, unsigned long long sr, unsigned long long sg,
                        unsigned long long sb) {
    if (a.accum != nullptr) {
        unsigned long long* acc = a.accum + static_cast<size_t>(local_pix) * 4u;
        if (a.sample_offset != 0u) {
            sr += acc[0];
            sg += acc[1];
            sb += acc[2];
        }
        acc[0] = sr;
        acc[1] = sg;
        acc[2] = sb;
    }
    return resolve_pixel(sr, sg, sb, a.sample_offset + a.spp, a.quantiser);
}

HDI uint32_t tile_global_row(uint32_t lr, uint32_t row_block, uint32_t rank, uint32_t count) {
    if (count <= 1) return lr;
    return ((lr / row_block) * count + rank) * row_block + lr % row_block;
}
// (Round 4 tried the quotients by the frame's constants -- width, row_block -- as multiply-high and shift with host-made magic numbers:
// two instructions instead of a dozen, four more kernel arguments -- and the cover frame 6.62 -> 6.76 ms: the clustered kernels have more
// wave-uniform values than scalar registers, and four more of them cost more v_readlanes than the divisions cost instructions.)
HDI uint32_t tile_global_row(const PathArgs& a, uint32_t lr) { return tile_global_row(lr, a.row_block, a.tile_rank, a.tile_count); }
HDI uint32_t pixel_row(const PathArgs& a, uint32_t pix) { return pix / a.width; }

// ============================================================================
// CH05 / CH06 — the reference's shaders
// ============================================================================

// GLSL dot/normalize without contraction (left-to-right)
DI float gdot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DI f3 gnormalize(f3 a) {
    const float l = __builtin_sqrtf(gdot(a, a));
    return mk(a.x / l, a.y / l, a.z / l);
}

// Correctly rounded square root and quotient WITHOUT the range handling hipcc wraps around them (LEAN instantiation).
// hipcc expands sqrtf(x) into v_sqrt_f32 (1 ulp) and a check of the two neighbouring floats against the residual --
// after multiplying x by 2^32 when it lies below 2^-96, and with a class test for 0 / inf / NaN at the end: 16
// instructions, of which the 10 of lean_sqrt are the core; it expands a / b into v_rcp_f32, one Newton step on the reciprocal
// and three on the quotient between v_div_scale_f32 (which rescales operands whose exponents are extreme or far apart),
// v_div_fmas_f32 and v_div_fixup_f32 (which put the scale back and deal with 0 / inf / NaN / denormals): 12 instructions,
// of which the 8 of lean_div are the core.  Where no rescaling happens the full forms ARE the cores, instruction for
// instruction, so the results are bit-identical: for x == 0 or x >= 2^-96, and for a == 0 or normal a, b with exponents
// less than 96 apart and a normal quotient.  launch_ch selects the LEAN kernels only for a UBO whose viewport and focal
// length lie in [2^-20, 2^20]: every square root the shaders take is then 0 or at least 2^-63 (a difference of two
// numbers of at least 2^-40 in magnitude, or a sum of squares), every divisor between 2^-21 and 2^22, every dividend
// 0 or at least 2^-44.  (One difference that cannot reach a pixel: -0 / b comes out as +0; each quotient of the shaders
// is added to 1 or compared with 0 next.)
DI float lean_sqrt(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = fma_(-s_dn, s, x), r_up = fma_(-s_up, s, x);
    const float r = (0.0f >= r_dn) ? s_dn : s;
    return (0.0f < r_up) ? s_up : r;
}
DI float lean_div(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float r1 = fma_(fma_(-b, r0, 1.0f), r0, r0);
    const float q0 = a * r1;
    const float q1 = fma_(fma_(-b, q0, a), r1, q0);
    return fma_(fma_(-b, q1, a), r1, q1);
}
// The correctly rounded square root in SIX instructions: v_rsq_f32 and one Newton step on the root, g = x y, h = y / 2, s = g + (x - g g) h.
// That the step lands on RN(sqrt(x)) is not a theorem about one-ulp starting values -- it is a fact about this GPU's v_rsq_f32, and a
// function of ONE float can be checked on every float: rtSelfTestUnaryScan (function 0) evaluates it on all 1 879 048 192 floats of [2^-96, 2^128)
// against sqrtf -- no mismatch (tests/test_gpu_parity.py; tools/sqrt_scan.hip tried five such forms, all exact there, this is the
// shortest).  The operand is kept away from zero by one v_max_f32, so that +0 gives +0 and -0 gives -0 (y = 2^63, g = +-0, s = +-0);
// NaN gives NaN (g is NaN).  NOT for negative operands (a negative number comes out where sqrtf gives NaN) nor +inf (NaN): the one
// site whose operand can be negative keeps lean_sqrt (psqrt_signed).  Below 2^-96 the residual loses bits to underflow and the result
// may be an ulp off, as lean_sqrt's may: it stays within a few ulps of sqrt(x), which is all the ray-sphere tests' argument below needs.
DI float newton_sqrt(float x) {
    const float y = __builtin_amdgcn_rsqf(__builtin_fmaxf(x, 0x1p-126f));
    const float g = x * y, h = 0.5f * y;
    return fma_(fma_(-g, g, x), h, g);
}
// ... and the correctly rounded reciprocal in THREE: v_rcp_f32 and one Newton step.  The same kind of fact, checked the same way:
// equal to 1.0f / s on every float of [2^-64, 2^64) (rtSelfTestUnaryScan, function 1; the compiler's core, lean_div(1, s), takes eight
// instructions).  unit3_scattered's 1 / sqrt(dot): the root lies in [1e-8, 4).
DI float newton_rcp(float s) {
    const float r = __builtin_amdgcn_rcpf(s);
    return fma_(fma_(-s, r, 1.0f), r, r);
}

// PATH mode takes the same lean forms wherever their precondition holds for EVERY input the kernels can see (psqrt, pdiv: 10 and 8
// instructions instead of 16 and 12; a segment takes a dozen square roots).  Site by site:
//   * the ray-sphere tests (disc = hb^2 - cc >= +0): lean_sqrt equals sqrtf for disc == 0 and disc >= 2^-96.  Below 2^-96 it may be
//     off, but any value it can return there lies in [0, 2^-47), like the exact root (< 2^-48), and the root -hb -+ sq is then the
//     same float either way: for |hb| >= 2^-22 both differ from -hb by less than half an ulp of hb (2^-46 at least) and round to -hb;
//     for |hb| < 2^-22 both roots are below 2^-21 < t_min = 0.001 and the sphere is no hit.  NaN, negative, -0 and +inf arguments come
//     out as from sqrtf (NaN, NaN, -0, +inf): the instruction's own results, which no comparison of the correction step replaces;
//   * random_unit_vector: 1 - z^2 with z = 1 - 2u, u a multiple of 2^-24 in [0, 1): 0 or at least 2^-23; the lens: sqrt(u), 0 or >= 2^-24;
//   * dielectric: 1 - cos^2 with cos <= 1 a float: 0, negative (-> NaN both ways) or a multiple of 2^-48 not below 2^-25;
//     |1 - dot(perp, perp)|: 1 - x is exact for a float x in [0.5, 2] and a multiple of 2^-24, below 0.5 it is >= 0.5;
//   * unit3 of a SCATTERED direction: lambertian directions have a component of at least 1e-8 in magnitude (the near-zero rule replaces
//     the others by the normal), so dot >= 1e-16 and its root lies in [1e-8, 4): divisor and dividend (1.0) are normal, 27 binades
//     apart at most; a metal or glass direction is a reflection or refraction of a unit vector (length 1 to a few ulps), a fuzzed one the
//     sum of a unit vector and fuzz <= 1 times another, which comes out exactly 0 (absorbed before unit3: dir . n > 0 fails) or with a
//     component of at least 2^-25 unless all three cancel to within 2^-48 of their operands at once -- 2^-72 of the fuzzed samples
//     would have to; such a sample may differ from the oracle in its last bits.  unit3 of a CAMERA ray kept hipcc's forms until late in
//     round 4 (its length is whatever the caller's camera makes it); rtRender now bounds that length (camera_rays_moderate);
//   * resolve_pixel: scale * sum with scale >= 2^-48 and sum an integer: 0 or >= 2^-48.
// Late in round 4 the square roots became newton_sqrt (six instructions, above) at every site but the dielectric's 1 - cos^2, the only
// operand that can be negative: for x == 0 or x >= 2^-96 it IS sqrtf(x), like lean_sqrt, so every argument above stands; the lock-step
// test of the large spheres takes the root of any discriminant and reads it only when the discriminant is >= +0.  The one quotient of
// the kernels, unit3_scattered's 1 / length, became newton_rcp (three instructions).
// -DRTIOW_LEAN_PATH=0: hipcc's forms everywhere; =1: lean_sqrt everywhere (A/B and parity checks of the above: the frames are the same).
#ifndef RTIOW_LEAN_PATH
#define RTIOW_LEAN_PATH 2
#endif
DI float psqrt(float x) { return RTIOW_LEAN_PATH == 2 ? newton_sqrt(x) : RTIOW_LEAN_PATH ? lean_sqrt(x) : __builtin_sqrtf(x); }
DI float psqrt_signed(float x) { return RTIOW_LEAN_PATH ? lean_sqrt(x) : __builtin_sqrtf(x); }  // NaN for a negative operand, as sqrtf
DI float precip(float s) { return RTIOW_LEAN_PATH == 2 ? newton_rcp(s) : RTIOW_LEAN_PATH ? lean_div(1.0f, s) : 1.0f / s; }  // unit3_scattered's 1 / length

#ifndef RTIOW_TU_PART
// ch_kernel_tiles: the shaders as the reference dispatches them -- 16x16 workgroup as raytrace06.comp:2, one lane per
// pixel; ceil-div grid with a bounds check (fixes the fixed 64x64 / floor-div dispatches of main.cpp:321 and RTCHAP05
// main.cpp:306).  One packed 32-bit store per lane: a wave writes four 64-byte row segments.  Kept for degenerate images
// (one row or one column: u or v is 0/0 there, and the row kernel below relies on them being finite) and as the
// statement-for-statement form the row kernel is checked against (tests: RtParams.kernel = 1 selects it).
__global__ __launch_bounds__(256) void ch_kernel_tiles(ChArgs a) {
    const uint32_t tiles_x = (a.width + 15u) / 16u;
    const uint32_t gx = (blockIdx.x % tiles_x) * 16u + (threadIdx.x & 15u);
    const uint32_t gy = (blockIdx.x / tiles_x) * 16u + (threadIdx.x >> 4);
    // u depends on the column only and v on the row only (raytrace06.comp:57-58): 32 IEEE divisions per
    // 16x16 tile instead of 512, shared through LDS; the quotients are the per-pixel ones, bit for bit
    __shared__ float uv[32];
    if (threadIdx.x < 16u)
        uv[threadIdx.x] = static_cast<float>((blockIdx.x % tiles_x) * 16u + threadIdx.x) / (a.ubo.imageWidth - 1);
    else if (threadIdx.x < 32u)
        uv[threadIdx.x] = static_cast<float>((blockIdx.x / tiles_x) * 16u + (threadIdx.x - 16u)) / (a.ubo.imageHeight - 1);
    __syncthreads();
    if (gx >= a.width || gy >= a.height) return;

    // raytrace06.comp:53-61
    const f3 origin = mk(0.0f, 0.0f, 0.0f);
    const f3 horizontal = mk(a.ubo.viewportWidth, 0.0f, 0.0f);
    const f3 vertical = mk(0.0f, a.ubo.viewportHeight, 0.0f);
    f3 llc;
    llc.x = ((origin.x - horizontal.x / 2) - vertical.x / 2) - 0.0f;
    llc.y = ((origin.y - horizontal.y / 2) - vertical.y / 2) - 0.0f;
    llc.z = ((origin.z - horizontal.z / 2) - vertical.z / 2) - a.ubo.focalLength;
    const float u = uv[threadIdx.x & 15u];
    const float v = uv[16u + (threadIdx.x >> 4)];
    f3 dir;
    dir.x = ((llc.x + horizontal.x * u) + vertical.x * v) - origin.x;
    dir.y = ((llc.y + horizontal.y * u) + vertical.y * v) - origin.y;
    dir.z = ((llc.z + horizontal.z * u) + vertical.z * v) - origin.z;

    // hitSphere: raytrace06.comp:21-33 / raytrace05.comp:21-30
    const f3 centre = mk(0.0f, 0.0f, -1.0f);
    const float radius = 0.5f;
    const f3 oc = mk(origin.x - centre.x, origin.y - centre.y, origin.z - centre.z);
    const float qa = gdot(dir, dir);
    const float qb = 2.0f * gdot(oc, dir);
    const float qc = gdot(oc, oc) - radius * radius;
    const float disc = qb * qb - 4 * qa * qc;

    f3 col;
    bool shaded = false;
    if (a.mode == RT_MODE_CH05) {
        if (disc > 0) {  // raytrace05.comp:29,35-37
            col = mk(1.0f, 0.0f, 0.0f);
            shaded = true;
        }
    } else {
        const float t = (disc < 0) ? -1.0f : (-qb - __builtin_sqrtf(disc)) / (2.0f * qa);
        if (t > 0.0f) {  // raytrace06.comp:39-43
            const f3 r = mk(origin.x + dir.x * t, origin.y + dir.y * t, origin.z + dir.z * t);
            const f3 nrm = gnormalize(mk(r.x - 0.0f, r.y - 0.0f, r.z - (-1.0f)));
            col = mk(0.5f * (nrm.x + 1), 0.5f * (nrm.y + 1), 0.5f * (nrm.z + 1));
            shaded = true;
        }
    }
    if (!shaded) {  // raytrace06.comp:45-47
        const f3 unit = gnormalize(dir);
        const float t = 0.5f * (unit.y + 1.0f);
        const float k = 1.0f - t;
        col = mk(1.0f * k + 0.5f * t, 1.0f * k + 0.7f * t, 1.0f * k + 1.0f * t);
    }
    // imageStore(vec4(color,0.0)) into rgba8: alpha byte 0 (raytrace06.comp:66)
    a.dst[static_cast<size_t>(gy) * a.dst_stride + gx] =
        pack_rgb(quant_unorm8(col.x), quant_unorm8(col.y), quant_unorm8(col.z));
}


// ch_kernel_rows: the same arithmetic, laid out for the memory system.  At 4 bytes per pixel the shaders should be
// bound by HBM writes, but one lane per pixel spends ~200 IEEE-exact instructions on each (round 2: 1.44 TB/s, 18 % of
// the 8 TB/s peak, at 16384^2).  Two observations cut that to ~60 for a sky pixel and ~100 for a sphere pixel without
// changing one rounding:
//   * raytrace06.comp:57-61 makes dir.x a function of the COLUMN only, dir.y of the ROW only and dir.z a constant
//     (horizontal = (w,0,0), vertical = (0,h,0): the cross terms are 0 * u and 0 * v, +0 for the finite, non-negative
//     u, v of an image with at least two rows and columns).  So dir.x, dir.x*dir.x and oc.x*dir.x are computed once per
//     column of a 256-column tile and dir.y, dir.y*dir.y, oc.y*dir.y once per row, by the very expressions of the
//     shader, and shared through LDS; per pixel, gdot(dir,dir) = (xx + yy) + zz is two additions of the SAME three
//     products the per-pixel form adds -- same operands, same order, same roundings;
//   * normalize(dir) is only ever read for its y component (raytrace06.comp:45-46): one division instead of three.
// A lane takes four consecutive pixels of a row and writes them with one 16-byte store: a wave writes 1 KB of one row
// (eight whole 128-byte lines), a workgroup 4 x rows_per_wave rows of the same 256 columns, whose per-column values
// stay in registers from row to row.
struct ChConst {  // per-frame constants of raytrace06.comp:53-56, 21-27, by the shader's own expressions
    float hx, vy, llc_x, llc_y, dz, zz, oz_dz, oc_x, oc_y, qc;
    float qb;  // LEAN kernels: 2 * gdot(oc, dir), the same for every pixel (see ch_pixel)
};
DI ChConst ch_constants(const RtUbo5& ubo) {
    const f3 origin = mk(0.0f, 0.0f, 0.0f);
    const f3 horizontal = mk(ubo.viewportWidth, 0.0f, 0.0f);
    const f3 vertical = mk(0.0f, ubo.viewportHeight, 0.0f);
    ChConst k;
    k.hx = horizontal.x;
    k.vy = vertical.y;
    k.llc_x = ((origin.x - horizontal.x / 2) - vertical.x / 2) - 0.0f;
    k.llc_y = ((origin.y - horizontal.y / 2) - vertical.y / 2) - 0.0f;
    const float llc_z = ((origin.z - horizontal.z / 2) - vertical.z / 2) - ubo.focalLength;
    // dir.z = ((llc.z + horizontal.z * u) + vertical.z * v) - origin.z with horizontal.z * u = vertical.z * v = +0
    k.dz = ((llc_z + 0.0f) + 0.0f) - origin.z;
    const f3 centre = mk(0.0f, 0.0f, -1.0f);
    const float radius = 0.5f;
    const f3 oc = mk(origin.x - centre.x, origin.y - centre.y, origin.z - centre.z);
    k.oc_x = oc.x;
    k.oc_y = oc.y;
    k.zz = k.dz * k.dz;
    k.oz_dz = oc.z * k.dz;
    k.qc = gdot(oc, oc) - radius * radius;
    k.qb = 2.0f * k.oz_dz;
    return k;
}

// quant_unorm8 in four instructions: (int)(clamp(x, 0, 1) * 255 + 0.5) == min(u32(x * 255 + 0.5), 255) for every float x,
// the conversion being v_cvt_u32_f32 -- truncation toward zero, negative values and NaN to 0, overflow to 2^32 - 1: below
// 0 the clamp gives 0.5 -> 0 and the conversion 0; inside [0, 1] the two expressions are the same float; above 1 the clamp
// gives 255.5 -> 255 and the conversion something >= 255, which the integer minimum brings back.  (Written as asm: a C cast
// of an out-of-range float is undefined, the instruction is not.)
// BOUNDED (the lean kernels): the caller's x is known to lie in [-2^-10, 1 + 2^-10] and the minimum is dropped -- x * 255 + 0.5 is
// then below 256.  That holds for every colour the shaders make from a moderate UBO: normalize() divides a component by the
// correctly rounded root of a sum of squares that contains its own square, so a unit vector's component is 1 + a few ulps at
// most in magnitude; 0.5 * (n + 1) and the sky's (1 - t) + c * t with t = 0.5 * (unit.y + 1), c <= 1 stay within a few ulps
// of [0, 1]; raytrace05's red is (1, 0, 0).
template <bool BOUNDED = false>
DI uint32_t ch_unorm8(float x) {
    const float y = x * 255.0f + 0.5f;
    uint32_t q;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(q) : "v"(y));
    return BOUNDED ? q : (q < 255u ? q : 255u);
}

// raytrace06.comp:41-43 from v = rayAt(t) - centre and l = sqrt(dot(v, v)): normalize(v) = v / l, colour 0.5 * (N + 1), rgba8
template <bool LEAN>
DI uint32_t ch_normal_colour(f3 v, float l) {
    auto div_ = [](float a, float b) { return LEAN ? lean_div(a, b) : a / b; };
    const f3 nrm = mk(div_(v.x, l), div_(v.y, l), div_(v.z, l));
    const f3 col = mk(0.5f * (nrm.x + 1), 0.5f * (nrm.y + 1), 0.5f * (nrm.z + 1));
    return pack_rgb(ch_unorm8<LEAN>(col.x), ch_unorm8<LEAN>(col.y), ch_unorm8<LEAN>(col.z));  // alpha byte 0 (raytrace06.comp:66)
}
// raytrace06.comp:45-47 from unit_y = normalize(dir).y, the only component of it the shaders read
template <bool LEAN>
DI uint32_t ch_sky_colour(float unit_y) {
    const float t = 0.5f * (unit_y + 1.0f);
    const float kk = 1.0f - t;
    const f3 col = mk(1.0f * kk + 0.5f * t, 1.0f * kk + 0.7f * t, 1.0f * kk + 1.0f * t);
    return pack_rgb(ch_unorm8<LEAN>(col.x), ch_unorm8<LEAN>(col.y), ch_unorm8<LEAN>(col.z));
}

// one pixel: raytrace06.comp:21-48 / raytrace05.comp:21-40 from the hoisted products (xx = dir.x*dir.x, ox = oc.x*dir.x, ...)
template <bool LEAN>
DI uint32_t ch_pixel(uint32_t mode, const ChConst& k, float dx, float xx, float ox, float dy, float yy, float oy) {
    auto sqrt_ = [](float x) { return LEAN ? lean_sqrt(x) : __builtin_sqrtf(x); };
    auto div_ = [](float a, float b) { return LEAN ? lean_div(a, b) : a / b; };
    const float qa = (xx + yy) + k.zz;             // gdot(dir, dir)
    // 2 * gdot(oc, dir).  oc = origin - centre = (+0, +0, 1) by the shader's constants, so oc.x * dir.x and oc.y * dir.y are zeros
    // of either sign for the finite dir.x, dir.y of a frame with finite UBO values, their sum is a zero, and a zero plus
    // oc.z * dir.z = -focalLength is that number exactly when it is not itself zero: under the LEAN kernels' precondition
    // (|focalLength| in [2^-20, 2^20]) the three-term sum IS k.oz_dz for every pixel, bit for bit, and is taken from the frame
    // constants -- five instructions and two LDS operands per pixel less.  (The full kernels, launched for any other UBO, add it up.)
    const float qb = LEAN ? k.qb : 2.0f * ((ox + oy) + k.oz_dz);
    const float disc = qb * qb - 4 * qa * k.qc;
    if (mode == RT_MODE_CH05) {
        if (disc > 0) return pack_rgb(ch_unorm8<LEAN>(1.0f), ch_unorm8<LEAN>(0.0f), ch_unorm8<LEAN>(0.0f));  // raytrace05.comp:29,35-37
    } else {
        const float t = (disc < 0) ? -1.0f : div_(-qb - sqrt_(disc), 2.0f * qa);
        if (t > 0.0f) {  // raytrace06.comp:39-43
            const f3 r = mk(0.0f + dx * t, 0.0f + dy * t, 0.0f + k.dz * t);
            const f3 v = mk(r.x - 0.0f, r.y - 0.0f, r.z - (-1.0f));
            return ch_normal_colour<LEAN>(v, sqrt_(gdot(v, v)));  // normalize(v) = v / sqrt(dot(v, v))
        }
    }
    return ch_sky_colour<LEAN>(div_(dy, sqrt_(qa)));  // raytrace06.comp:45-47: only the y component of normalize(dir) is used
}

// ---- Two-phase pixels: the bytes of the frame without every rounding of the way there ---------------------------------------------
// What the shaders store is 24 bits per pixel, and both of their colour formulas end in a quantiser that forgets almost everything
// the correctly rounded roots and quotients before it were careful about.  The two-phase form of ch_kernel_rows (Ziv's strategy, as
// correctly rounded math libraries use it) computes each pixel from one-ulp hardware approximations first, together with a proof
// that the quantised result cannot depend on the difference, and falls back to the exact arithmetic above -- ch_pixel<true>'s own
// functions -- for the pixels where that proof does not hold (about 1 in 10^4 sky pixels, 1 in 1400 sphere pixels).  Same bytes:
//   * SKY.  The colour is a function F of the single float unit_y = RN(dy / RN(sqrt(qa))) (ch_sky_colour), a step function with 280
//     steps in 179 zones a few ulps wide; tools/gen_ch_sky_table.py finds every one of them by evaluating the shader's arithmetic on
//     all 25 million values unit_y + 1 can take, and writes rtiow_ch_sky_table.h: per bucket of 1/256 of unit_y one zone [lo, hi]
//     widened by G = 2^-21 either side and the colours below and above it.  Phase 1: y1 = dy * v_rsq_f32(qa).  With r = dy / sqrt(qa)
//     (|r| <= 1 + 2^-22: qa contains dy * dy) and h = 2^-24, unit_y = r (1 + e2) / (1 + e1) with |e1|, |e2| <= h lies within 2 h |r|
//     (1 + h) of r, and y1 = r (1 + e6)(1 + e7) with |e6| <= 2 h (one ulp: tests/test_gpu_ch_two_phase.py measures it on every float of a
//     binade pair), |e7| <= h within 3 h |r| (1 + h): |y1 - unit_y| < 5.01 h < 0.63 G.  So y1 < lo means unit_y lies below the zone's
//     first step, y1 > hi means it lies at or above the last, and F(unit_y) is the table's colour; anything else is computed.
//   * SPHERE (raytrace06's normal colour).  t is exact as before (it is multiplied into three coordinates, and its error would be
//     amplified 250-fold on the way to a byte); v and d = dot(v, v) are the shader's.  Per channel, rho = v_c / sqrt(d) (|rho| <= 1 +
//     2^-22) and Y = 127.5 rho + 128.  The shader's y = RN(RN(0.5 RN(n + 1) * 255) + 0.5) with n = RN(v_c / RN(sqrt(d))): |n - rho| <=
//     2 h |rho| (1 + h), RN(n + 1) adds at most 2 h, the product and the sum at most 2^-17 each (y < 256):  |y - Y| <= 127.5 * 4.01 h +
//     2^-16 < 4.6e-5.  Phase 1: y1 = fma(v_c * v_rsq_f32(d), 127.5, 128): |y1 - Y| <= 127.5 * 3.01 h + 2^-17 < 3.1e-5.  The byte is
//     trunc(y) (0.49 < y < 255.51): when fract(y1) lies in [D, 1 - D] with D = 2^-13 = 1.22e-4 > 1.6 * 7.7e-5, y and y1 lie between the same
//     two integers.  Otherwise the pixel's three channels are divided out exactly.
// The table sits in LDS (8 KB per workgroup, one ds_read_b128 per pixel, neighbours read the same entry).
#include "rtiow_ch_sky_table.h"
alignas(16) __device__ const uint32_t ch_sky_table_words[4u * RTIOW_CH_SKY_BUCKETS] = RTIOW_CH_SKY_TABLE_WORDS;
constexpr float kChSkyScale = 256.0f;          // bucket = trunc(fma(y1, 256, 256)): 0 .. 512
constexpr float kChSphereGuard = 0x1p-13f;     // D
constexpr uint32_t kChPhase2Flag = 1u << 31;   // (rtSelfTestArith ops 11, 14, 16 report the phase in the alpha byte's top bit)

DI float ch_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
DI uint32_t ch_cvt_u32(float y) {
    uint32_t q;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(q) : "v"(y));
    return q;
}
// phase 1 of the sky: the table's colour for y1; `second`: y1 lies within G of a step and the pixel has to be computed.  `table`: the
// 513 entries, in LDS for the kernel.  (An index past the table cannot happen for finite y1 <= 1 + 2^-21; LDS reads out of range return 0
// on this hardware.)
DI uint32_t ch_sky_phase1(const uint4* table, float y1, bool& second) {
    const uint4 e = table[ch_cvt_u32(fma_(y1, kChSkyScale, kChSkyScale))];
    const bool not_below = y1 >= __uint_as_float(e.x);
    second = not_below && y1 <= __uint_as_float(e.y);
    return not_below ? e.w : e.z;
}
DI uint32_t ch_sky_two_phase(const uint4* table, float dy, float qa) {
    bool second;
    const uint32_t c = ch_sky_phase1(table, dy * ch_rsq(qa), second);
    if (__builtin_expect(!second, 1)) return c;
    return ch_sky_colour<true>(lean_div(dy, lean_sqrt(qa))) | kChPhase2Flag;
}
DI uint32_t ch_normal_two_phase(f3 v) {
    const float d = gdot(v, v);
    const float inv = ch_rsq(d);
    const float yx = fma_(v.x * inv, 127.5f, 128.0f), yy = fma_(v.y * inv, 127.5f, 128.0f), yz = fma_(v.z * inv, 127.5f, 128.0f);
    const float fx = __builtin_amdgcn_fractf(yx) - 0.5f, fy = __builtin_amdgcn_fractf(yy) - 0.5f, fz = __builtin_amdgcn_fractf(yz) - 0.5f;
    const float worst = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fx), __builtin_fabsf(fy)), __builtin_fabsf(fz));
    if (__builtin_expect(worst <= 0.5f - kChSphereGuard, 1)) return pack_rgb(ch_cvt_u32(yx), ch_cvt_u32(yy), ch_cvt_u32(yz));
    return ch_normal_colour<true>(v, lean_sqrt(d)) | kChPhase2Flag;
}
// Is the pixel the sphere's (raytrace05.comp:29 disc > 0; raytrace06.comp:28,39 !(disc < 0), then t > 0 -- see below)?  disc = RN(qb qb -
// RN(4 qa qc)): the sign of a float difference is the sign of the exact one, so the product m = RN(qa * (4 qc)) (4 qc = 3 exactly, and
// RN(4 qa * qc) = RN(qa * 3): a power of two changes no rounding) is compared with qb * qb without being subtracted -- m < qb qb for
// CH05, m <= qb qb for CH06, which for the finite positive numbers of the lean range is m < the float after qb qb: one comparison
// against a per-frame bound either way.
DI float ch_sphere_bound(uint32_t mode, const ChConst& k) {
    const float qbqb = k.qb * k.qb;
    return mode == RT_MODE_CH05 ? qbqb : __uint_as_float(__float_as_uint(qbqb) + 1u);
}
DI bool ch_is_sphere(const ChConst& k, float bound, float qa) { return qa * (4 * k.qc) < bound; }
// the sphere's colour, two-phase (the pixel is known to be the sphere's by ch_is_sphere).  raytrace06.comp:39 also asks for t > 0:
// with disc >= 0 that is -qb > sqrt(disc), true for every such pixel when focalLength > 0 (qb = -2 focalLength and disc <= qb^2 / 4)
// and for none when it is negative (the sphere is behind the camera: the whole frame is sky, these pixels by the exact arithmetic).
DI uint32_t ch_sphere_two_phase(uint32_t mode, const ChConst& k, float dx, float dy, float qa) {
    if (mode == RT_MODE_CH05) return pack_rgb(255u, 0u, 0u);  // = ch_unorm8 of (1, 0, 0)
    const float disc = k.qb * k.qb - qa * (4 * k.qc);  // = qb * qb - 4 * qa * qc, the product as ch_is_sphere has it (see there)
    const float t = lean_div(-k.qb - newton_sqrt(disc), 2.0f * qa);  // (disc is +0 or at least 2^-63 here: the exact root)
    if (__builtin_expect(!(t > 0.0f), 0)) return ch_sky_colour<true>(lean_div(dy, lean_sqrt(qa))) | kChPhase2Flag;
    // rayAt's "origin + t * dir" adds +0 to each product in the shader; that turns a -0 into +0 and nothing else, and a zero coordinate of
    // either sign squares to +0 and divides to a zero that is 0.5 two operations later: the additions are left out (the second phase is
    // ch_normal_colour on the same v: equal wherever v is not a zero, and the bytes are equal there too)
    return ch_normal_two_phase(mk(dx * t, dy * t, k.dz * t - (-1.0f)));
}
// A lane's four pixels of a row, two-phase (LEAN preconditions: launch_ch).  The sky's first phase runs for all four side by side, without
// a branch (one wait for four table reads), whenever a lane of the wave has a sky pixel among them; sphere pixels and second phases follow
// under branches that most waves skip.  (The sphere's first phase side by side for waves that are all sphere: bit-identical, and slower --
// 0.17 against 0.145 ms at 16384 x 8192; hipcc keeps the four chains one after the other and materialises the flags in registers.)
// Returns the four pixels, alpha 0.
DI uint4 ch_quad_two_phase(uint32_t mode, const ChConst& k, const uint4* table, float4 dx4, float4 xx4, float dy, float yy) {
    const float dx[4] = {dx4.x, dx4.y, dx4.z, dx4.w}, xx[4] = {xx4.x, xx4.y, xx4.z, xx4.w};
    const float bound = ch_sphere_bound(mode, k);
    float qa[4];
    bool sphere[4], second[4] = {false, false, false, false};
    uint32_t px[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        qa[i] = (xx[i] + yy) + k.zz;
        sphere[i] = ch_is_sphere(k, bound, qa[i]);
    }
    const bool all_sphere = sphere[0] && sphere[1] && sphere[2] && sphere[3], any_sphere = sphere[0] || sphere[1] || sphere[2] || sphere[3];
    if (__builtin_amdgcn_ballot_w64(!all_sphere) != 0ull) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            px[i] = ch_sky_phase1(table, dy * ch_rsq(qa[i]), second[i]);
            second[i] = second[i] && !sphere[i];
        }
        if (__builtin_amdgcn_ballot_w64(second[0] || second[1] || second[2] || second[3]) != 0ull) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (second[i]) px[i] = ch_sky_colour<true>(lean_div(dy, lean_sqrt(qa[i])));
        }
    }
    if (__builtin_amdgcn_ballot_w64(any_sphere) != 0ull) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (sphere[i]) px[i] = ch_sphere_two_phase(mode, k, dx[i], dy, qa[i]) & ~kChPhase2Flag;
    }
    return make_uint4(px[0], px[1], px[2], px[3]);
}

constexpr uint32_t kChTileCols = 256;  // columns of a workgroup's tile: 64 lanes x 4 pixels
enum : int { kChFull = 0, kChLean = 1, kChTwoPhase = 2 };  // hipcc's roots and quotients / their cores / two-phase pixels
template <int FORM>
__global__ __launch_bounds__(256) void ch_kernel_rows(ChArgs a) {
    __shared__ float4 col_dx[kChTileCols / 4], col_xx[kChTileCols / 4], col_ox[kChTileCols / 4];  // per column, four to a lane
    __shared__ float row_dy[64], row_yy[64], row_oy[64];                                          // per row of the tile
    const ChConst k = ch_constants(a.ubo);
    const uint32_t tiles_x = (a.width + kChTileCols - 1u) / kChTileCols;
    const uint32_t rows_per_block = 4u * a.rows_per_wave;  // (<= 64)
    const uint32_t col0 = (blockIdx.x % tiles_x) * kChTileCols, row0 = (blockIdx.x / tiles_x) * rows_per_block;
    {   // raytrace06.comp:57,59: u and dir.x of this thread's column (0 * v = +0: v is finite and >= 0)
        const float u = static_cast<float>(col0 + threadIdx.x) / (a.ubo.imageWidth - 1);
        const float dx = ((k.llc_x + k.hx * u) + 0.0f) - 0.0f;
        reinterpret_cast<float*>(col_dx)[threadIdx.x] = dx;
        reinterpret_cast<float*>(col_xx)[threadIdx.x] = dx * dx;
        if (FORM == kChFull) reinterpret_cast<float*>(col_ox)[threadIdx.x] = k.oc_x * dx;
    }
    if (threadIdx.x < rows_per_block) {  // raytrace06.comp:58,60: v and dir.y of the tile's rows (0 * u = +0)
        const float v = static_cast<float>(row0 + threadIdx.x) / (a.ubo.imageHeight - 1);
        const float dy = ((k.llc_y + 0.0f) + k.vy * v) - 0.0f;
        row_dy[threadIdx.x] = dy;
        row_yy[threadIdx.x] = dy * dy;
        if (FORM == kChFull) row_oy[threadIdx.x] = k.oc_y * dy;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t c = col0 + 4u * lane;
    if (c >= a.width) return;
    const float4 dx4 = col_dx[lane], xx4 = col_xx[lane];
    const float4 ox4 = FORM == kChFull ? col_ox[lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const bool vec = c + 3u < a.width && a.vector_store != 0u;  // a whole, 16-byte aligned quad
    for (uint32_t j = 0; j < a.rows_per_wave; ++j) {
        const uint32_t lr = wave * a.rows_per_wave + j, row = row0 + lr;
        if (row >= a.height) break;
        const float dy = row_dy[lr], yy = row_yy[lr], oy = FORM == kChFull ? row_oy[lr] : 0.0f;  // LDS broadcast
        uint4 px;
        px.x = ch_pixel<FORM == kChLean>(a.mode, k, dx4.x, xx4.x, ox4.x, dy, yy, oy);
        px.y = ch_pixel<FORM == kChLean>(a.mode, k, dx4.y, xx4.y, ox4.y, dy, yy, oy);
        px.z = ch_pixel<FORM == kChLean>(a.mode, k, dx4.z, xx4.z, ox4.z, dy, yy, oy);
        px.w = ch_pixel<FORM == kChLean>(a.mode, k, dx4.w, xx4.w, ox4.w, dy, yy, oy);
        uint32_t* out = a.dst + static_cast<size_t>(row) * a.dst_stride + c;
        if (vec) {
            *reinterpret_cast<uint4*>(out) = px;
        } else {  // the ragged right edge, or a destination that is not 16-byte aligned
            out[0] = px.x;
            if (c + 1u < a.width) out[1] = px.y;
            if (c + 2u < a.width) out[2] = px.z;
            if (c + 3u < a.width) out[3] = px.w;
        }
    }
}

// ch_kernel_rows<kChTwoPhase>: the two-phase pixels.  With a dozen instructions per sky pixel the kernel is bound by how its stores reach
// HBM where the frame is sky and by the vector ALU where it is the sphere, and the sphere sits in the middle of the frame: dealt in raster
// order, workgroups of the same kind run at the same time (and, tile columns recurring with the period of the dispatcher's round over XCDs
// and CUs, on the same CUs).  So the tiles are dealt in an order that mixes them: row block rb of the grid renders row block rb * stride mod
// n (stride near the golden section of n, coprime with it) and starts three tile columns further right than the block before it
// (tools/ch_bandwidth.py, 16384^2 / 16384 x 8192: 0.231 / 0.162 ms in raster order, 0.212 / 0.145 so).  Per-column values come from
// LDS as in the other forms (u and v by the lean quotient: launch_ch checks their operands too), each wave computes those of its own rows.
// (Sixteen waves around one copy of the table instead of four: slower at every size, 0.237 against 0.222 ms at 16384^2.)
template <>
__global__ __launch_bounds__(256) void ch_kernel_rows<kChTwoPhase>(ChArgs a) {
    __shared__ uint4 sky_table[RTIOW_CH_SKY_BUCKETS];
    __shared__ float4 col_dx[kChTileCols / 4], col_xx[kChTileCols / 4];  // per column of the tile, four to a lane
    __shared__ float row_dy[4][16], row_yy[4][16];  // per wave: dir.y and its square for the wave's rows (rows_per_wave <= 16)
    const ChConst k = ch_constants(a.ubo);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tiles_x = (a.width + kChTileCols - 1u) / kChTileCols;
    const uint32_t grid_row = blockIdx.x / tiles_x;
    const uint32_t rb = static_cast<uint32_t>((static_cast<unsigned long long>(grid_row) * a.row_block_stride) % a.row_blocks);
    const uint32_t col0 = ((blockIdx.x % tiles_x + 3u * grid_row) % tiles_x) * kChTileCols;
    const uint32_t row0 = (rb * 4u + wave) * a.rows_per_wave;
    for (uint32_t i = threadIdx.x; i < RTIOW_CH_SKY_BUCKETS; i += 256u) sky_table[i] = reinterpret_cast<const uint4*>(ch_sky_table_words)[i];
    if (threadIdx.x < kChTileCols) {  // raytrace06.comp:57,59: u and dir.x of this thread's column (0 * v = +0: v is finite and >= 0)
        const float u = lean_div(static_cast<float>(col0 + threadIdx.x), a.ubo.imageWidth - 1);
        const float dx = ((k.llc_x + k.hx * u) + 0.0f) - 0.0f;
        reinterpret_cast<float*>(col_dx)[threadIdx.x] = dx;
        reinterpret_cast<float*>(col_xx)[threadIdx.x] = dx * dx;
    }
    if (lane < a.rows_per_wave) {  // raytrace06.comp:58,60: v and dir.y of this wave's rows (0 * u = +0)
        const float v = lean_div(static_cast<float>(row0 + lane), a.ubo.imageHeight - 1);
        const float dy = ((k.llc_y + 0.0f) + k.vy * v) - 0.0f;
        row_dy[wave][lane] = dy;
        row_yy[wave][lane] = dy * dy;
    }
    __syncthreads();
    const uint32_t c = col0 + 4u * lane;
    if (c >= a.width) return;
    const bool vec = c + 3u < a.width && a.vector_store != 0u;  // a whole, 16-byte aligned quad
    const float4 dx4 = col_dx[lane], xx4 = col_xx[lane];
    for (uint32_t j = 0; j < a.rows_per_wave; ++j) {
        const uint32_t row = row0 + j;
        if (row >= a.height) break;
        const uint4 px = ch_quad_two_phase(a.mode, k, sky_table, dx4, xx4, row_dy[wave][j], row_yy[wave][j]);
        uint32_t* out = a.dst + static_cast<size_t>(row) * a.dst_stride + c;
        if (vec) {
            *reinterpret_cast<uint4*>(out) = px;
        } else {  // the ragged right edge, or a destination that is not 16-byte aligned
            out[0] = px.x;
            if (c + 1u < a.width) out[1] = px.y;
            if (c + 2u < a.width) out[2] = px.z;
            if (c + 3u < a.width) out[3] = px.w;
        }
    }
}

#endif  // RTIOW_TU_PART

// ============================================================================
// PATH mode building blocks (BUILD-SPEC: SURVEY.md section 9)
// ============================================================================

struct Path {
    f3 o, du, att;  // origin, unit direction, attenuation product
    Pcg rng;
    DI Path() : rng(0u) {}
};

// Rejection-free sampling.  A rejection loop on a 64-wide SIMD runs for the MAXIMUM trip
// count over the lanes, so the spec uses closed forms made of + - * sqrt and fma only.
// sincos_2pi: quadrant from the top two bits of u, then two fma-only polynomials in
// f = frac(4u) for sin(pi/2 f) and cos(pi/2 f) (max abs error 2e-7).
DI void sincos_2pi(float u, float& c, float& s) {
    constexpr float S0 = 0x1.921fb6p+0f, S1 = -0x1.4abbc4p-1f, S2 = 0x1.4668f0p-4f, S3 = -0x1.32533cp-8f,
                    S4 = 0x1.3e15f6p-13f;
    constexpr float C0 = 0x1.fffffep-1f, C1 = -0x1.3bd3a6p+0f, C2 = 0x1.03bd02p-2f, C3 = -0x1.54f5dcp-6f,
                    C4 = 0x1.c1ecap-11f;
    const float t = 4.0f * u;
    const int q = static_cast<int>(t);
    const float f = t - static_cast<float>(q);
    const float f2 = f * f;
    const float sp = f * fma_(f2, fma_(f2, fma_(f2, fma_(f2, S4, S3), S2), S1), S0);
    const float cp = fma_(f2, fma_(f2, fma_(f2, fma_(f2, C4, C3), C2), C1), C0);
    // q: 0 (c,s)=(cp,sp)  1 (-sp,cp)  2 (-cp,-sp)  3 (sp,-cp)
    const float a = (q & 1) ? sp : cp;
    const float b = (q & 1) ? cp : sp;
    c = (q == 1 || q == 2) ? -a : a;
    s = (q >= 2) ? -b : b;
}

DI f3 random_unit_vector(Pcg& rng) {
    const float u1 = rng.uniform();
    const float u2 = rng.uniform();
    const float z = 1.0f - 2.0f * u1;
    const float r = psqrt(fma_(-z, z, 1.0f));
    float c, s;
    sincos_2pi(u2, c, s);
    return mk(r * c, r * s, z);
}

// one camera sample of pixel (i, j): SURVEY 9.4 / 9.5
DI void camera_path(const PathArgs& a, uint32_t i, uint32_t j, uint32_t sample, Path& p) {
    // (The camera's 22 floats stay kernel arguments in SGPRs, although the clustered kernels then have half again as many wave-uniform
    // values as scalar registers and keep the rest in VGPR lanes, a v_readlane away.  Round 2 read them from the kernel-argument segment
    // here instead: scalar loads, 9.6 -> 10.25 ms.  Round 4 read them from a copy in LDS -- five broadcast reads per pass, 50 fewer
    // spilled SGPRs: -0.3 ... -0.7 %, but the values are then per-lane registers in the kernel's tightest spot and 28 registers of other
    // wave-uniform state go to scratch; dropped for the variant that spills nothing.  What did help the scalar side: every per-wave
    // LDS area at a fixed offset from one base instead of seven bases -- 690 -> 490 v_readlanes in the code, no time either way.)
    const RtCamera& c = a.cam;
    p.rng = Pcg(a.seed, j * a.width + i, sample);
    const float u = (static_cast<float>(i) + p.rng.uniform()) * a.inv_wm1;
    const float v = (static_cast<float>(j) + p.rng.uniform()) * a.inv_hm1;
    f3 off = mk(0.0f, 0.0f, 0.0f);
    if (c.lens_radius > 0.0f) {  // wave-uniform; random_in_unit_disk: radius sqrt(u1), azimuth 2 pi u2
        const float u1 = p.rng.uniform();
        const float u2 = p.rng.uniform();
        const float r = psqrt(u1);
        float cs, sn;
        sincos_2pi(u2, cs, sn);
        const float rdx = c.lens_radius * (r * cs), rdy = c.lens_radius * (r * sn);
        off = mk(fma_(c.v[0], rdy, c.u[0] * rdx), fma_(c.v[1], rdy, c.u[1] * rdx),
                 fma_(c.v[2], rdy, c.u[2] * rdx));
    }
    p.o = mk(c.origin[0] + off.x, c.origin[1] + off.y, c.origin[2] + off.z);
    f3 d;
    d.x = fma_(v, c.vertical[0], fma_(u, c.horizontal[0], c.lower_left[0])) - c.origin[0] - off.x;
    d.y = fma_(v, c.vertical[1], fma_(u, c.horizontal[1], c.lower_left[1])) - c.origin[1] - off.y;
    d.z = fma_(v, c.vertical[2], fma_(u, c.horizontal[2], c.lower_left[2])) - c.origin[2] - off.z;
    // (rtRender refuses a camera whose rays could be shorter than 2^-30 or longer than 2^40 -- camera_rays_moderate, rtiow_device.h --
    // so the ray's length is inside the short root's and reciprocal's domains like that of a scattered ray)
    p.du = RTIOW_LEAN_PATH == 2 ? unit3_scattered(d) : unit3(d);
    p.att = mk(1.0f, 1.0f, 1.0f);
}

DI f3 sky_radiance(const Path& p) {  // raytrace06.comp:45-47, direction already unit
    const float t = 0.5f * (p.du.y + 1.0f);
    const float k = 1.0f - t;
    const f3 c = mk(fma_(t, 0.5f, k), fma_(t, 0.7f, k), fma_(t, 1.0f, k));
    return mk(p.att.x * c.x, p.att.y * c.y, p.att.z * c.z);
}

// Hit at distance s on a sphere with centre `ctr` and shading record `m`: scatter per
// material (metal fuzz as book v4: reflected + fuzz * random_unit_vector()).
// Returns false when the path is absorbed (radiance 0).
DI bool scatter(f3 ctr, const ShadeRec& m, float s, Path& p) {
    const f3 hit = mk(fma_(s, p.du.x, p.o.x), fma_(s, p.du.y, p.o.y), fma_(s, p.du.z, p.o.z));
    // inv_r = 1/radius (rounded once, on the host); negative radius flips the normal
    const f3 outward = mk((hit.x - ctr.x) * m.inv_r, (hit.y - ctr.y) * m.inv_r, (hit.z - ctr.z) * m.inv_r);
    const float dn = dot3(p.du, outward);
    const bool front = dn < 0.0f;
    const f3 n = front ? outward : mk(-outward.x, -outward.y, -outward.z);
    f3 dir;
    // lambertian and fuzzy metal both start with one random unit vector: drawn once for the lanes of
    // either kind (in a wave all material branches are taken anyway, so this halves that code)
    f3 rv = mk(0.0f, 0.0f, 0.0f);
    if (m.kind == RT_MAT_LAMBERTIAN || (m.kind == RT_MAT_METAL && m.param > 0.0f)) rv = random_unit_vector(p.rng);
    if (m.kind == RT_MAT_LAMBERTIAN) {
        dir = mk(n.x + rv.x, n.y + rv.y, n.z + rv.z);
        if (__builtin_fabsf(dir.x) < 1e-8f && __builtin_fabsf(dir.y) < 1e-8f &&
            __builtin_fabsf(dir.z) < 1e-8f)
            dir = n;
        p.att = mk(p.att.x * m.albedo[0], p.att.y * m.albedo[1], p.att.z * m.albedo[2]);
    } else if (m.kind == RT_MAT_METAL) {
        const float k2 = 2.0f * dot3(p.du, n);
        const f3 refl = mk(fma_(-k2, n.x, p.du.x), fma_(-k2, n.y, p.du.y), fma_(-k2, n.z, p.du.z));
        dir = refl;
        if (m.param > 0.0f)  // fuzz
            dir = mk(fma_(m.param, rv.x, refl.x), fma_(m.param, rv.y, refl.y), fma_(m.param, rv.z, refl.z));
        if (!(dot3(dir, n) > 0.0f)) return false;
        p.att = mk(p.att.x * m.albedo[0], p.att.y * m.albedo[1], p.att.z * m.albedo[2]);
    } else {
        // (a glass record's albedo words hold 1 / ior and Schlick's quotient for either side: rtSetScene)
#ifndef RTIOW_GLASS_HOST
#define RTIOW_GLASS_HOST 1  // (-DRTIOW_GLASS_HOST=0: A/B only -- the two quotients in the kernel)
#endif
        const float ratio = front ? (RTIOW_GLASS_HOST ? m.albedo[0] : 1.0f / m.param) : m.param;  // 1 / ior : ior
        const float nd = -dot3(p.du, n);
        const float cosv = (nd < 1.0f) ? nd : 1.0f;
        const float sinv = psqrt_signed(fma_(-cosv, cosv, 1.0f));  // (cosv < -1 by an ulp: negative, NaN as in the oracle)
        bool reflect = ratio * sinv > 1.0f;
        if (!reflect) {  // Schlick; (1-cos)^5 by repeated multiply, never powf
            float r0 = RTIOW_GLASS_HOST ? (front ? m.albedo[1] : m.albedo[2]) : (1.0f - ratio) / (1.0f + ratio);
            r0 = r0 * r0;
            const float x = 1.0f - cosv;
            const float x2 = x * x;
            const float x5 = (x2 * x2) * x;
            const float prob = fma_(1.0f - r0, x5, r0);
            reflect = prob > p.rng.uniform();
        }
        if (reflect) {
            const float k2 = 2.0f * dot3(p.du, n);
            dir = mk(fma_(-k2, n.x, p.du.x), fma_(-k2, n.y, p.du.y), fma_(-k2, n.z, p.du.z));
        } else {
            const f3 perp = mk(ratio * fma_(cosv, n.x, p.du.x), ratio * fma_(cosv, n.y, p.du.y),
                               ratio * fma_(cosv, n.z, p.du.z));
            const float par = -psqrt(__builtin_fabsf(1.0f - dot3(perp, perp)));
            dir = mk(fma_(par, n.x, perp.x), fma_(par, n.y, perp.y), fma_(par, n.z, perp.z));
        }
    }
    p.o = hit;
    p.du = unit3_scattered(dir);
    return true;
}

// Stage the sphere list into LDS as {cx, cy, cz, r*r}: 16 B per sphere
// (485 -> 7.6 KiB, 4096 -> 64 KiB; gfx950 has 160 KiB per CU).  Padding entries
// {0,0,0,-1} can never be hit: disc = hb^2 - |o|^2 - 1 < 0.
DI void stage_spheres(const PathArgs& a, float4* lds, uint32_t n_pad) {
    for (uint32_t i = threadIdx.x; i < n_pad; i += blockDim.x) {
        float4 s = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
        if (i < a.n) {
            s = a.spheres[i];
            s.w = s.w * s.w;
        }
        lds[i] = s;
    }
}

// Closest hit, straightforward form: all lanes walk the LDS list in lock-step
// (same address in every lane: LDS broadcast read, no bank conflicts).
DI int closest_hit_simple(const float4* lds, uint32_t n, const Path& p, float& best) {
    best = __builtin_inff();
    int best_i = -1;
    for (uint32_t i = 0; i < n; ++i) {
        const float4 s = lds[i];
        const float ocx = p.o.x - s.x, ocy = p.o.y - s.y, ocz = p.o.z - s.z;
        const float hb = fma_(ocz, p.du.z, fma_(ocy, p.du.y, ocx * p.du.x));
        const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
        const float disc = fma_(hb, hb, -cc);
        if (__builtin_signbit(disc) || disc != disc) continue;
        const float sq = psqrt(disc);
        float root = -hb - sq;
        if (!(root > kTMin && root < best)) {
            root = -hb + sq;
            if (!(root > kTMin && root < best)) continue;
        }
        best = root;
        best_i = static_cast<int>(i);
    }
    return best_i;
}

#ifndef RTIOW_TU_PART
// ============================================================================
// PATH v1: one lane per pixel (reference form; kept as a cross-check and ablation)
// ============================================================================
__global__ __launch_bounds__(256) void path_pixel_kernel(PathArgs a) {
    extern __shared__ float4 lds_spheres[];
    __shared__ unsigned int blk_paths, blk_segments;
    if (threadIdx.x == 0) {
        blk_paths = 0;
        blk_segments = 0;
    }
    stage_spheres(a, lds_spheres, a.n);
    __syncthreads();

    const uint32_t tiles_x = (a.width + 15u) / 16u;
    const uint32_t i = (blockIdx.x % tiles_x) * 16u + (threadIdx.x & 15u);
    const uint32_t lr = (blockIdx.x / tiles_x) * 16u + (threadIdx.x >> 4);
    uint32_t n_paths = 0, n_segments = 0;
    if (i < a.width && lr < a.local_rows) {
        const uint32_t j = tile_global_row(a, lr);
        unsigned long long sr = 0ull, sg = 0ull, sb = 0ull;
        for (uint32_t s = 0; s < a.spp; ++s) {
            Path p;
            camera_path(a, i, j, a.sample_offset + s, p);
            ++n_paths;
            for (uint32_t depth = 0; depth < a.max_depth; ++depth) {
                ++n_segments;
                float dist;
                const int idx = closest_hit_simple(lds_spheres, a.n, p, dist);
                if (idx < 0) {
                    const f3 rad = sky_radiance(p);
                    sr += to_fixed(rad.x);
                    sg += to_fixed(rad.y);
                    sb += to_fixed(rad.z);
                    break;
                }
                const float4 g = lds_spheres[idx];
                if (!scatter(mk(g.x, g.y, g.z), a.shade[idx], dist, p)) break;
            }
        }
        a.dst[static_cast<size_t>(lr) * a.dst_stride + i] = close_pixel(a, lr * a.width + i, sr, sg, sb);
    }
    atomicAdd(&blk_paths, n_paths);
    atomicAdd(&blk_segments, n_segments);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&a.counters->paths, static_cast<unsigned long long>(blk_paths));
        atomicAdd(&a.counters->segments, static_cast<unsigned long long>(blk_segments));
    }
}

#endif  // RTIOW_TU_PART

// ============================================================================
// PATH v2: persistent waves, per-pixel LDS accumulators, ballot refill, candidate bitmasks
// ============================================================================
//
// Every lane owns kSlots path slots; one loop iteration traces exactly one
// segment per live slot.
//
//   refill   two-level queue.  Eight global queues, one per XCD, each head on its
//            own cache line, hand out pools of consecutive pixels (one atomicAdd
//            per wave per pool; a line saturates near 90 atomics/us, far below
//            one dequeue per lane).
//            A wave walks its pool pixel by pixel, sample by sample: the idle
//            slots of the wave are found with __ballot, ranked with mbcnt and
//            given consecutive samples of the current pixel, no memory traffic.
//            Dead lanes are refilled at once, so the sphere loop always runs
//            with full waves; pools shrink to one pixel as the queue drains.
//            (ACCEL: the samples go to the primary pass instead -- up to 64 camera rays, one per lane, are made,
//            traced against the spheres their pixels' cones reach and shaded on the spot; the paths that go on
//            reach the idle slots through LDS records.  See "The primary pass".)
//   trace    all lanes walk the LDS sphere list in lock-step (broadcast reads),
//            branch-free: the sign bit of each discriminant is shifted into a
//            per-lane candidate word by one v_alignbit.  Only candidates (about
//            two per ray) take the sqrt/root path, per lane, after each block of
//            32 spheres.  (ACCEL: the two-level clustered list instead, see
//            trace_clustered; the rest of the kernel is shared.)
//   sparse   path lengths are heavy-tailed (mean 2.8 segments, 0.1 % reach depth 50
//            inside glass; seen from the slots, where a path counts for as long as it
//            lives, one slot in fifty holds such a path), so once the queue is empty
//            EVERY wave is left with a handful of long paths, and a lock-step iteration
//            costs the same for 1 live lane as for 128.  What then sets the end of the
//            frame is the latency of one iteration, forty to fifty times over.  With few
//            live paths the wave turns the loop inside out.  Flat list (trace_sparse): for
//            each live path in turn the 64 LANES take 64 different spheres per step and a
//            6-step wave reduction on the (root, index) key finds the hit.  Clustered list
//            (trace_sparse_parallel): all live paths at once, the lanes sharing out the
//            (path, box) and (path, member) pairs through LDS work lists; the paths are
//            first gathered in slot 0, so that shading runs once per iteration, not once
//            per slot.  Same minimum, same tie rule as the lock-step pass.
//            (Round 1 also pooled the leftovers of a workgroup's waves in one wave; with the
//            path-parallel trace that costs more than it saves -- a lone wave's iteration is
//            no dearer for running beside three others' -- and was removed: cover frame
//            10.73 -> 10.33 ms, one eighth of it 1.99 -> 1.75 ms.)
//   shade    miss -> sky radiance, converted to fixed point and added to the
//            pixel's LDS accumulator (ds_add_u64; integer sums do not depend on
//            order); hit -> scatter by material (shading records in LDS too
//            while the list is small).
//   resolve  every pixel in flight owns one of the wave's 64 LDS accumulator
//            entries {r, g, b, samples done}.  The lane whose sample completes
//            the pixel (returning ds_add on the counter) converts it to RGBA8
//            and stores it; pixels of a wave finish in pixel order, so the L2
//            merges the 4-byte stores into full lines.  This store is the only
//            HBM traffic of the frame besides the one-time scene read: no
//            accumulation buffer, no global atomics, no resolve pass.
//
// The closest hit is order-independent: it is the minimum over spheres of each
// sphere's first root in (t_min, inf), ties to the lowest index — exactly what
// the oracle's sequential scan computes.

// Samples a wave takes from its queue at a time.  A fetch costs two dependent atomics on the queue's
// head, microseconds when many waves share it, during which the wave's idle lanes wait: a pool must
// last a few iterations; too large and the end of the frame balances badly.  The eight heads sit on
// separate 128-byte lines (rtiow_device.h).  While they shared one line -- one L2 channel serialising
// every fetch of the chip -- small pools were ruinous (cover frame, clustered list: 128 samples -> 35.0
// ms, 384 -> 14.2, 1536 -> 11.4; flat list: 100 -> 36.3, 384 -> 29.8) and the sweep over all eight
// drying queues at the end of a frame stalled every wave for ~180 us.  With separate lines
// (tools/ab_bench.py, clustered list, full frame / one eighth of it): 1700 samples 10.97 / 3.00 ms,
// 850 -> 10.48 / 2.29, 450 -> 10.46 / 2.05, 256 -> 10.5 / 1.95; flat list 29.5 / 5.4 from 256 to 450.
// launch_path takes kPoolWork / (tests a segment costs) samples, at least 256.
// Round 4: where a wave's share of the frame is large (8192 samples and more: launch_path) pools are twice that -- the cover frame: nine
// pixels instead of four -- and never more than an EIGHTH of the share (a quarter elsewhere, as before).  Since the default kernel went to sixteen waves per CU the cover frame has fewer than 256 pixels per
// wave and hands out no whole chunks any more -- every pixel came from a four-pixel pool, 240 000 queue fetches a frame and, with the
// dear head of each queue dealt pixel by pixel (which only small frames need: launch_path), 330 000: at ~40 bytes of fabric traffic
// each they WERE the frame's HBM traffic (15.5 MB against 3.84 MB of pixels: profiles/r04_traffic_experiments.txt), and the waits
// cost 1.3 % besides: cover frame 6.55 -> 6.47 ms, 2 / 4 / 8 tiles and 16 / 500 spp unchanged (profiles/r04_pool_rule.txt).
// Tried on top of this and dropped (each cost the full frame 2-7 %): asking for the next pool ahead of
// time (one atomic in flight per wave); lanes 0-7 looking at all eight heads in one round trip, on every
// fetch or only once the wave's own queue is dry (the queue-by-queue sweep at the very end of a frame
// costs a wave ~65 us, but eight loads per fetch on contended lines cost more).
#ifndef RTIOW_POOL_WORK
#define RTIOW_POOL_WORK 40000u
#endif
constexpr uint32_t kPoolWork = RTIOW_POOL_WORK;
#ifndef RTIOW_LONG_FROM
#define RTIOW_LONG_FROM 20  // (round 2: 12; re-tuned for sixteen waves per CU: 8 / 12 / 16 / 20 / 32 -> cover frame 7.28 / 7.21 / 7.23 / 7.16 / 7.35 ms,
                            // one eighth of it 1.26 / 1.26 / 1.26 / 1.24 / 1.25)
#endif
#ifndef RTIOW_LONG_WEIGHT
#define RTIOW_LONG_WEIGHT 128
#endif
#ifndef RTIOW_COST_SAMPLE
#define RTIOW_COST_SAMPLE 16  // of the pixels handed out one by one, every n-th reports its cost for its neighbours too (a power of two).
                              // (Rounds 2-3: every fourth.  A report is a memory-side atomic, ~40 bytes of fabric traffic; since the cover frame hands
                              // out no whole chunks -- section "Stores" -- a reporting frame made 240 000 of them, 9.6 MB.  Two reports per 32-pixel
                              // chunk order the next frame as well as eight: frames standing still 6.196 / 6.190 / 6.200 ms with every 4th / 8th / 16th,
                              // an orbit +7.4 / +7.8 / +6.5 % -- profiles/r04_moving_camera.txt -- and a path of more than kLongFrom segments reports
                              // for itself whatever its pixel.)
#endif
constexpr uint32_t kCostSample = RTIOW_COST_SAMPLE;
constexpr uint32_t kLongFrom = RTIOW_LONG_FROM, kLongWeight = RTIOW_LONG_WEIGHT;  // cost of a path of more segments than kLongFrom, for the chunk order
#ifndef RTIOW_FAR_LOCKSTEP
#define RTIOW_FAR_LOCKSTEP 1  // (-DRTIOW_FAR_LOCKSTEP=0: A/B only -- a ray from outside the boxes' range takes every cluster of a one-level scene)
#endif
#ifndef RTIOW_SLOTS
#define RTIOW_SLOTS 2  // (-DRTIOW_SLOTS=1: ablation only -- half the paths in flight)
#endif
constexpr int kSlots = RTIOW_SLOTS;  // path slots per lane
static_assert(kSlots == 1 || kSlots == 2, "the work lists encode the path slot in one bit");
constexpr uint32_t kBlockSph = 32;  // spheres per candidate word
#ifndef RTIOW_SPARSE_MAX
#define RTIOW_SPARSE_MAX 16
#endif
constexpr uint32_t kSparseMax = RTIOW_SPARSE_MAX; // live paths per wave at or below which the sphere-parallel trace runs
#ifndef RTIOW_SPARSE_MAX_ACCEL
#define RTIOW_SPARSE_MAX_ACCEL 32  // all paths together (trace_sparse_parallel)
#endif
constexpr uint32_t kSparseMaxAccel = RTIOW_SPARSE_MAX_ACCEL;  // the same for the clustered list (cluster-parallel trace)

// Bookkeeping of a path in flight, ONE register: its pixel's accumulator entry (one of its wave's kAccEntries; the field has 10 bits),
// the line buffer of its chunk + 1 (0: the pixel goes straight to the frame: 3 bits) and the segments it has taken
// (19 bits: rtRender caps max_depth accordingly).  The pixel itself is looked up by the entry when it completes (lds_pix: one word per
// accumulator entry, written when the pixel is opened).  Rounds 1-3 carried pixel, entry, line and depth in a register each, eight
// per lane -- the first place to look when the four-waves variant spilled 28.
constexpr uint32_t kMetaLineShift = 10, kMetaDepthShift = 13;
constexpr uint32_t kPixLineShift = 29;  // lds_pix: the entry's pixel (rtRender caps a tile at 2^29 pixels) | its line buffer + 1 << 29
#ifndef RTIOW_RESOLVE_BATCH
#define RTIOW_RESOLVE_BATCH 16  // (4 / 8 / 16: cover frame 6.18 / 6.17 / 6.14 ms against 6.21 resolving at once; 64 entries per wave)
#endif
constexpr uint32_t kResolveBatch = RTIOW_RESOLVE_BATCH;  // done entries a wave lets gather before it resolves them (resolve_done)
static_assert(kMaxPathDepth == (1u << (32u - kMetaDepthShift)) - 1u, "rtRender's cap on max_depth is the depth field of Slot::meta");
DI uint32_t meta_of(uint32_t entry, uint32_t line) { return entry | (line << kMetaLineShift); }
DI uint32_t meta_entry(uint32_t m) { return m & ((1u << kMetaLineShift) - 1u); }
[[maybe_unused]] DI uint32_t meta_line(uint32_t m) { return (m >> kMetaLineShift) & 7u; }
DI uint32_t meta_depth(uint32_t m) { return m >> kMetaDepthShift; }
struct Slot {
    Path p;
    uint32_t meta;
    bool active;
};

}  // namespace (PersistArgs is shared with the second compilation of this file: see the end of it)
struct PersistArgs {
    uint32_t n_pad;        // slots of the LDS sphere list: the flat list padded to a multiple of four, or the clustered list's slots
    uint32_t total_pix;    // pixels of this tile (the global queue counts pixels)
    uint32_t total_waves;  // waves of the grid
    uint32_t pool_pix;     // pixels per pool away from the tail
    uint32_t chunk_pool;   // pixels per pool while whole chunks are handed out (a multiple of kChunkPix)
    uint32_t chunk_until;  // ... which lasts while a queue has at least this many pixels left
    uint32_t fine_until;   // the first pixels of a queue's small-pool part that go out fine_pix at a time (0: none)
    uint32_t fine_pix;     // ... one iteration's worth of samples: 128 / spp pixels, at least one
    uint32_t wave_bytes;   // LDS a wave has to itself behind the scene (accumulators, pixels, line buffers, lists, records)
    // the primary pass of the clustered kernels (see "The primary pass" below)
    uint32_t use_pass;     // != 0: camera rays take the primary pass (enough samples per pixel), 0: straight into the slots
    uint32_t pass_keep;    // camera paths a wave may keep in LDS beyond its idle slots (records of its own; 0: none)
    uint32_t pass_min_idle;  // idle slots a wave with live paths must have before it runs a pass (kPassMinLanes less the records)
    uint32_t pass_cap;     // camera paths a pass may make at most (<= 64: one per lane, and the records must hold them)
    uint32_t primary_all;  // != 0: no cone cull (degenerate image size, or a lens that leaves the boxes' range)
    float lens_rho;        // bound of the lens offset |off|, with its margin
    float h_len, v_len;    // |cam.horizontal|, |cam.vertical|, with their margin
    float abs_margin;      // absolute slack of the cone test: 2^-16 of the scene's coordinate range
};
using PersistentKernelFn = void (*)(PathArgs, PersistArgs);
PersistentKernelFn small_clustered_kernel(bool flat);  // path_persistent_kernel<true, true, flat>, from the second compilation of this file
PersistentKernelFn large_clustered_kernel(bool flat);  // path_persistent_kernel<false, true, flat>, from the third
namespace {

// ---- Cold kernel arguments are re-read where they are used (round 5) ----------------------------------------------------------
// A persistent kernel holds every argument it touches in a scalar register from its first instruction to its last: the compiler
// loads the by-value structs once (22 s_load at the top of the default kernel) and never again.  PathArgs + PersistArgs are ~100
// dwords, the loop state another ~30, a wave has 102 SGPRs: round 4's default kernel kept 199 scalar values in VGPR lanes, and
// tools/blockprof counted what that costs -- 3.1e8 v_readlane / v_writelane per cover frame, 6.4 % of all vector instructions
// (profiles/r05_instruction_mix_base.txt).  Most of those arguments are COLD: the camera's 22 floats and the cone margins are used
// once per primary pass (1.7 million a frame, ~1000 instructions each), the queue's constants once per pixel opened, the frame's
// pointers once per batch of resolved pixels.  reload_*() reads them again from the kernel-argument segment -- scalar loads, served
// by the scalar cache -- through a pointer the compiler cannot see through (the empty asm), so that the loads stay where they are
// written and the values die with the stage that uses them.  `a` and `g` themselves are only to be used for what the per-segment
// code needs (the list's shape, the boxes' range, spp, max_depth).  (Round 2 tried "the camera read from the kernel-argument segment
// at its use" with volatile loads inside camera_path: 9.6 -> 10.25 ms; that was 22 separate dependent loads in the middle of the
// camera code.  Here one batch of wide loads is issued at the head of a stage and waited for once.)
typedef const __attribute__((address_space(4))) unsigned char* KernargBytes;
DI KernargBytes kernarg_opaque() {
    KernargBytes k = (KernargBytes)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}
constexpr size_t kPersistArgsOffset = (sizeof(PathArgs) + alignof(PersistArgs) - 1u) / alignof(PersistArgs) * alignof(PersistArgs);
// (the copies come from the constant address space with their structs' alignment -- the empty asm hides the segment's -- or the loads
// would be vector loads of the same address in every lane: s_load needs dword alignment)
DI PathArgs reload_path_args() {  // (only the fields the caller reads are loaded: the copy is scalarised)
    PathArgs r;
    __builtin_memcpy(&r, (const __attribute__((address_space(4))) PathArgs*)kernarg_opaque(), sizeof(PathArgs));
    return r;
}
DI PersistArgs reload_persist_args() {
    PersistArgs r;
    __builtin_memcpy(&r, (const __attribute__((address_space(4))) PersistArgs*)(kernarg_opaque() + kPersistArgsOffset), sizeof(PersistArgs));
    return r;
}

DI void examine_candidate(const float4* lds, uint32_t j, uint32_t n, const Path& p, float& best,
                          int& best_i) {
    if (j >= n) return;  // list padding
    const float4 s = lds[j];
    const float ocx = p.o.x - s.x, ocy = p.o.y - s.y, ocz = p.o.z - s.z;
    const float hb = fma_(ocz, p.du.z, fma_(ocy, p.du.y, ocx * p.du.x));
    const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
    const float disc = fma_(hb, hb, -cc);
    if (__builtin_signbit(disc) || disc != disc) return;
    const float sq = psqrt(disc);
    float root = -hb - sq;
    if (!(root > kTMin && root < best)) {
        root = -hb + sq;
        if (!(root > kTMin && root < best)) return;
    }
    // ascending j within a slot: a tie keeps the earlier (lower) index
    best = root;
    best_i = static_cast<int>(j);
}

// -DRTIOW_DEBUG_TIMELINE: three wall-clock stamps per wave (queue dry, first sparse iteration, end) binned into
// Counters::tl_hist -- cheap enough to leave the kernel's timing as it is (RTIOW_DEBUG_HIST=1 prints them).
#ifdef RTIOW_DEBUG_TIMELINE
#define TL_MARK(var) do { if ((var) == 0ull) (var) = wall_clock64(); } while (0)
#else
#define TL_MARK(var) ((void)0)
#endif
#ifdef RTIOW_DEBUG_COUNTERS
#define DBG_ADD(var, x) (var) += (x)
#define DBG_STAMP() __builtin_readcyclecounter()
#else
#define DBG_ADD(var, x) ((void)0)
#define DBG_STAMP() 0ull
#endif

template <int R>
DI void trace_slots(const float4* lds, uint32_t n_pad, uint32_t n, Slot (&sl)[R], float (&best)[R],
                    int (&best_i)[R], uint32_t& dbg_slow_trips, uint32_t& dbg_cands,
                    unsigned long long& dbg_t_slow) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        best[r] = __builtin_inff();
        best_i[r] = -1;
    }
    for (uint32_t base = 0; base < n_pad; base += kBlockSph) {
        uint32_t miss[R];
#pragma unroll
        for (int r = 0; r < R; ++r) miss[r] = 0u;
        // (the list is padded to a multiple of FOUR, not of the candidate word's 32: round 5 -- BASELINE config 2's four spheres took 32
        // tests per ray and slot, 41 % of that frame's vector instructions for padding)
        const uint32_t jn = n_pad - base < kBlockSph ? n_pad - base : kBlockSph;
#pragma unroll 4
        for (uint32_t j = 0; j < jn; ++j) {
            const float4 s = lds[base + j];  // wave-uniform address: LDS broadcast
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const Path& p = sl[r].p;
                const float ocx = p.o.x - s.x, ocy = p.o.y - s.y, ocz = p.o.z - s.z;
                const float hb = fma_(ocz, p.du.z, fma_(ocy, p.du.y, ocx * p.du.x));
                const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
                const float disc = fma_(hb, hb, -cc);
                // shift the sign bit in: 1 = certainly no hit
                miss[r] = __builtin_amdgcn_alignbit(miss[r], __float_as_uint(disc), 31);
            }
        }
        [[maybe_unused]] const unsigned long long ts0 = DBG_STAMP();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t cand = sl[r].active ? ~miss[r] << (kBlockSph - jn) : 0u;  // first sphere of the block at bit 31
            DBG_ADD(dbg_cands, __builtin_popcount(cand));
#ifdef RTIOW_DEBUG_COUNTERS
            {  // wave-level trip count of the loop below = max over lanes of popcount
                uint32_t m = __builtin_popcount(cand);
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t o = __shfl_xor(m, off);
                    m = o > m ? o : m;
                }
                if ((threadIdx.x & 63u) == 0u) dbg_slow_trips += m;
            }
#endif
            while (cand) {
                const uint32_t bit = static_cast<uint32_t>(__builtin_clz(cand));
                cand &= ~(0x80000000u >> bit);
                examine_candidate(lds, base + bit, n, sl[r].p, best[r], best_i[r]);
            }
        }
        DBG_ADD(dbg_t_slow, DBG_STAMP() - ts0);
    }
}

// Sphere-parallel closest hit for a wave with few live paths: see "sparse" above.  Each lane scans
// the spheres lane, lane+64, ... in ascending order with the oracle's per-sphere rule (near root if
// it lies beyond t_min, else the far root) and a strict '<', so within a lane a tie keeps the lower
// index; the wave reduction takes the minimum of (root bits << 32 | index), which orders positive
// floats numerically and breaks ties towards the lower index as well.
// `list` is the flat list (idx_map == nullptr: slot == sphere index) or the clustered list's slots
// (idx_map gives the original index, 0xFFFF for padding).  Results: best_i = slot, best_o = index.
template <int R>
DI void trace_sparse(const float4* list, const SlotIndex* idx_map, uint32_t n_slots, uint32_t n, Slot (&sl)[R],
                     float (&best)[R], int (&best_i)[R], uint32_t (&best_o)[R]) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        best[r] = __builtin_inff();
        best_i[r] = -1;
        best_o[r] = 0u;
        unsigned long long live = __ballot(sl[r].active);
        while (live != 0ull) {
            const int src = __builtin_ctzll(live);
            live &= live - 1ull;
            const Path& p = sl[r].p;
            const float ox = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p.o.x), src));
            const float oy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p.o.y), src));
            const float oz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p.o.z), src));
            const float dx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p.du.x), src));
            const float dy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p.du.y), src));
            const float dz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p.du.z), src));
            unsigned long long key = ~0ull;  // (root bits << 32) | original index
            uint32_t kslot = 0u;
            for (uint32_t j = lane; j < n_slots; j += 64u) {
                const float4 s = list[j];
                const uint32_t orig = idx_map ? idx_map[j] : j;
                const float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
                const float hb = fma_(ocz, dz, fma_(ocy, dy, ocx * dx));
                const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
                const float disc = fma_(hb, hb, -cc);
                const bool cand = !__builtin_signbit(disc) && disc == disc && orig < n;
                const float sq = psqrt(cand ? disc : 0.0f);
                float root = -hb - sq;
                root = root > kTMin ? root : -hb + sq;
                const unsigned long long k2 = (static_cast<unsigned long long>(__float_as_uint(root)) << 32) | orig;
                if (cand && root > kTMin && k2 < key) {
                    key = k2;
                    kslot = j;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long other = __shfl_xor(key, off);
                const uint32_t oslot = __shfl_xor(kslot, off);
                if (other < key) {
                    key = other;
                    kslot = oslot;
                }
            }
            if (lane == static_cast<uint32_t>(src) && key != ~0ull) {
                best[r] = __uint_as_float(static_cast<uint32_t>(key >> 32));
                best_i[r] = static_cast<int>(kslot);
                best_o[r] = static_cast<uint32_t>(key);
            }
        }
    }
}

// Two-level closest hit (SURVEY f-4).  Phase 0, wave in lock-step: the few large spheres, exactly as
// the flat list does it.  Phase 1, wave in lock-step: the box of every cluster against every live ray
// (LDS broadcast; slab test on centre + half extent with the ray's reciprocal direction), one candidate
// bit per cluster.  Phase 2, lane by lane: the 16 members of each candidate cluster; lanes read their own
// cluster, rotated by the lane number so that the 16 lanes of an LDS access group touch 16 different
// bank quads whatever clusters they are on.  Members go through the same discriminant / candidate-word /
// first-root-beyond-t_min rule as the flat list, and the hit is the minimum of
// (root bits << 32 | original index): the same sphere the flat scan returns, ties included.
// The slab test only has to be conservative (rtiow_clusters.cpp sizes the boxes for its rounding and
// the exact test's), so it may use v_rcp_f32 and any operation order; it never decides a hit.
// key = root bits << 32 | original index << 16 | slot (both below 65536: rtSetScene caps the scene)
DI void examine_keyed(const float4* slots, const SlotIndex* idx_map, uint32_t slot, const float ox, const float oy,
                      const float oz, const float dx, const float dy, const float dz, unsigned long long& key) {
    const float4 s = slots[slot];
    const float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
    const float hb = fma_(ocz, dz, fma_(ocy, dy, ocx * dx));
    const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
    const float disc = fma_(hb, hb, -cc);
    if (__builtin_signbit(disc) || disc != disc) return;
    const float sq = psqrt(disc);
    float root = -hb - sq;
    root = root > kTMin ? root : -hb + sq;
    if (!(root > kTMin)) return;
    // (Padding slots never get here: their r^2 is -inf, so cc = +inf and the discriminant is -inf or NaN
    // for every ray.  A finite r^2 would not do: with r^2 = -1, far from the origin a ray aimed at it has
    // hb^2 above |o|^2 + 1 through the rounding of |d|^2 alone -- tools/fuzz_kernels.py case 100069.)
    const uint32_t orig = idx_map[slot];
    const unsigned long long k2 =
        (static_cast<unsigned long long>(__float_as_uint(root)) << 32) | (orig << 16) | slot;
    key = k2 < key ? k2 : key;
}

// the 16 members of the cluster whose slots start at `base`, for one ray
DI void examine_cluster(const float4* slots, const SlotIndex* idx_map, uint32_t base, uint32_t lane, const float ox,
                        const float oy, const float oz, const float dx, const float dy, const float dz,
                        unsigned long long& key) {
    static_assert(kClusterSize == 16u && kClusterStride == 16u, "the member order below is an XOR on four bits of a cluster's first slot");
    uint32_t mm = 0u;
    // Member step k of a lane reads slot (base + rot) ^ k, rot = lane % 16: for every k the sixteen lanes of an LDS access group touch
    // sixteen different bank quads whatever clusters they are on, as with the rotation (k + rot) % 16 of rounds 1-3 -- but that one made
    // sixteen lane constants, which the compiler kept in sixteen registers all kernel long; this one is one XOR with an immediate on
    // the item's own number (`base` is a multiple of 16: the large spheres' slots are padded to whole clusters).
    const uint32_t first = base + (lane & (kClusterSize - 1u));
    // (round 5: the XOR is taken on the BYTE offset, hidden from the compiler, which otherwise turns (first << 4) ^ (k << 4) back into
    // (first ^ k) << 4 -- an XOR and a shift per member, 32 instructions a round where 15 XORs do: tools/blockprof, block of line 1258)
    uint32_t first_bytes = first * static_cast<uint32_t>(sizeof(float4));
    asm volatile("" : "+v"(first_bytes));
    // four members per step, their reads issued together (left to itself the compiler keeps two reads in flight and
    // waits for each after eleven instructions; an LDS read with per-lane addresses takes longer than that)
#pragma unroll
    for (uint32_t k0 = 0; k0 < kClusterSize; k0 += 4u) {
        float4 s4[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u)
            s4[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(slots) + (first_bytes ^ ((k0 + u) * static_cast<uint32_t>(sizeof(float4)))));
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const float4 s = s4[u];
            const float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
            const float hb = fma_(ocz, dz, fma_(ocy, dy, ocx * dx));
            const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
            const float disc = fma_(hb, hb, -cc);
            mm = __builtin_amdgcn_alignbit(mm, __float_as_uint(disc), 31);
        }
    }
    uint32_t cand = ~mm & (0xFFFFFFFFu >> (32u - kClusterSize));  // bit kClusterSize-1-k: member step k
    while (cand) {
        const uint32_t k = static_cast<uint32_t>(__builtin_clz(cand)) - (32u - kClusterSize);
        cand &= ~((1u << (kClusterSize - 1u)) >> k);
        examine_keyed(slots, idx_map, first ^ k, ox, oy, oz, dx, dy, dz, key);
    }
}

// Inclusive prefix sum over the 64 lanes on the DPP network (no LDS round trips): Hillis-Steele inside
// each row of 16 lanes, then the last lane of row 0/2 into row 1/3 and of row 1 into rows 2-3.
DI uint32_t wave_inclusive_sum(uint32_t v) {
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, true));  // row_shr:1
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, true));  // row_shr:2
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, true));  // row_shr:4
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, true));  // row_shr:8
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false)); // row_bcast:15
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false)); // row_bcast:31
    return v;
}

// Square root, reciprocal square root and reciprocal for the CULLS of the primary pass (cone_of_span, cone_reaches,
// cone_reaches_sphere): on the device the one-instruction forms (v_sqrt_f32, v_rsq_f32, v_rcp_f32: 1 ulp), on the host the
// exact ones.  The culls carry relative margins of 2^-6 .. 2^-8 and an absolute one of 2^-16 of the scene's range; an ulp
// is 2^-24.  (The correctly rounded sqrt and division cost 12 and 11 instructions each: 53 per pass saved.)  cull_sqrt
// is nudged up by two ulps so that it never falls below the exact root.
HDI float cull_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RTIOW_EXACT_CULL)  // (-DRTIOW_EXACT_CULL: A/B only)
    return __builtin_amdgcn_sqrtf(x) * 1.00000024f;
#else
    return __builtin_sqrtf(x);
#endif
}
HDI float cull_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RTIOW_EXACT_CULL)  // (-DRTIOW_EXACT_CULL: A/B only)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / __builtin_sqrtf(x);
#endif
}
HDI float cull_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RTIOW_EXACT_CULL)  // (-DRTIOW_EXACT_CULL: A/B only)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

HDI float slab_rcp(float d) {  // reciprocal of a direction component kept away from zero (finite slabs, no NaN)
    const float c = __builtin_fabsf(d) < 1e-18f ? __builtin_copysignf(1e-18f, d) : d;
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_rcpf(c);  // (v_rcp_f32, 1 ulp: the tests it feeds only cull, with margins far above that)
#else
    return 1.0f / c;
#endif
}

// Per-wave LDS of the clustered trace: the per-ray result keys, the (ray, cluster) work list of phase 2
// and, for scenes with super-clusters, the (ray, super-cluster) list before it.
constexpr uint32_t kLargeLockstep = 8;  // up to this many large spheres take the exact test in lock-step (trace_clustered)

// makes the fourth component of a float4 read from LDS count as used, so that the read stays one ds_read_b128
DI void keep_b128(const float4& v) { asm volatile("" ::"v"(v.w)); }

// slab test of one box (centre, half extent) against a ray given by 1/d, -o/d: sign bit of the result
// set = the ray leaves the box before it enters it (or before its origin)
HDI float slab_gap(const float4& mid, const float4& half, float ix, float iy, float iz, float ax, float ay, float az) {
    const float tcx = fma_(mid.x, ix, ax);
    const float tcy = fma_(mid.y, iy, ay);
    const float tcz = fma_(mid.z, iz, az);
    const float jx = __builtin_fabsf(ix), jy = __builtin_fabsf(iy), jz = __builtin_fabsf(iz);
    // entry: latest of the three near planes and the origin; exit: earliest far plane
    const float tn = __builtin_fmaxf(__builtin_fmaxf(fma_(-half.x, jx, tcx), fma_(-half.y, jy, tcy)),
                                     __builtin_fmaxf(fma_(-half.z, jz, tcz), 0.0f));
    const float tf = __builtin_fminf(__builtin_fminf(fma_(half.x, jx, tcx), fma_(half.y, jy, tcy)),
                                     fma_(half.z, jz, tcz));
    return tf - tn;
}

// The same with the flat axis taken out (rtiow_clusters.cpp): every cluster box spans the common interval flat_mid +-
// flat_half along one axis, so the ray's entry into and exit from THAT slab (tn_f >= 0, tf_f) are worked out once per
// ray, and a box is one float4 {mid a, mid b, half a, half b} of the other two axes: 6 fma, v_max3, v_min3, one
// subtraction -- 9 instructions and one 16-byte LDS read per ray and box instead of 13 and two.
HDI float slab_gap_flat(const float4& box, float ia, float ib, float oa, float ob, float tn_f, float tf_f) {
    const float tca = fma_(box.x, ia, oa);
    const float tcb = fma_(box.y, ib, ob);
    const float ja = __builtin_fabsf(ia), jb = __builtin_fabsf(ib);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(fma_(-box.z, ja, tca), fma_(-box.w, jb, tcb)), tn_f);
    const float tf = __builtin_fminf(__builtin_fminf(fma_(box.z, ja, tca), fma_(box.w, jb, tcb)), tf_f);
    return tf - tn;
}
// a ray's part in it: the two remaining axes' reciprocal direction and -o/d, and the ray's interval in the common slab
struct FlatRay {
    float ia, ib, oa, ob, tn_f, tf_f;
};
DI FlatRay flat_ray(const PathArgs& a, float ix, float iy, float iz, float ax, float ay, float az, float extra = 0.0f) {
    const uint32_t fa = a.flat_axis;  // (wave-uniform)
    const float i_f = fa == 0u ? ix : (fa == 1u ? iy : iz), o_f = fa == 0u ? ax : (fa == 1u ? ay : az);
    FlatRay f;
    f.ia = fa == 0u ? iy : ix;
    f.oa = fa == 0u ? ay : ax;
    f.ib = fa == 2u ? iy : iz;
    f.ob = fa == 2u ? ay : az;
    const float tc = fma_(a.flat_mid, i_f, o_f), jf = __builtin_fabsf(i_f);
    const float fh = a.flat_half + extra;  // (extra: far_box_margin, for a ray that starts outside the boxes' range)
    f.tn_f = __builtin_fmaxf(fma_(-fh, jf, tc), 0.0f);
    f.tf_f = fma_(fh, jf, tc);
    return f;
}

// What to add to every half extent of a box (and of the common interval of a flat scene) for a ray that starts at `o`, anywhere:
// rtiow_clusters.cpp, far_k / far_c (the root is v_sqrt_f32's, 1 ulp: the factors carry 0.1 % of slack).
DI float far_box_margin(const PathArgs& a, float ox, float oy, float oz) {
    const float qx = ox - a.ccenter[0], qy = oy - a.ccenter[1], qz = oz - a.ccenter[2];
    return fma_(a.cfar_k, __builtin_amdgcn_sqrtf(fma_(qz, qz, fma_(qy, qy, qx * qx))), a.cfar_c);
}

// does the ray reach box c (of `bounds`) enlarged by `extra`?  (the slab test of the lock-step stages, for one box)
template <bool FLAT>
DI bool reaches_enlarged(const float4* bounds, const PathArgs& a, uint32_t c, float ox, float oy, float oz, float dx, float dy, float dz,
                         float extra) {
    const float ix = slab_rcp(dx), iy = slab_rcp(dy), iz = slab_rcp(dz);
    const float ax = -ox * ix, ay = -oy * iy, az = -oz * iz;
    if (FLAT) {
        const FlatRay f = flat_ray(a, ix, iy, iz, ax, ay, az, extra);
        float4 b = bounds[c];
        b.z += extra;
        b.w += extra;
        return !__builtin_signbit(slab_gap_flat(b, f.ia, f.ib, f.oa, f.ob, f.tn_f, f.tf_f));
    }
    const float4 mid = bounds[2u * c];
    float4 half = bounds[2u * c + 1u];
    half.x += extra;
    half.y += extra;
    half.z += extra;
    return !__builtin_signbit(slab_gap(mid, half, ix, iy, iz, ax, ay, az));
}

// Box number i of the clustered list (clusters first, then super-clusters) as centre + half extent.  !FLAT: as stored.
// FLAT: the list holds the boxes without their flat axis only -- 16 bytes a box instead of 32 -- and the whole box is
// made up with the common interval, i.e. a box that CONTAINS the real one: fine for every test that only culls (the
// cone tests of the primary pass, the slab tests of the sparse trace).
template <bool FLAT>
DI void load_box(const float4* bounds, const PathArgs& a, uint32_t i, float4& mid, float4& half) {
    if (!FLAT) {
        mid = bounds[2u * i];
        half = bounds[2u * i + 1u];
        return;
    }
    const float4 b = bounds[i];
    const uint32_t fa = a.flat_axis;  // (wave-uniform)
    mid = make_float4(fa == 0u ? a.flat_mid : b.x, fa == 0u ? b.x : (fa == 1u ? a.flat_mid : b.y), fa == 2u ? a.flat_mid : b.y, 0.0f);
    half = make_float4(fa == 0u ? a.flat_half : b.z, fa == 0u ? b.z : (fa == 1u ? a.flat_half : b.w), fa == 2u ? a.flat_half : b.w, 0.0f);
}

// the ray of path slot (item bit 6) of lane (item bits 0-5), for every lane's own item
template <int R>
DI void fetch_item_ray(const Slot (&sl)[R], uint32_t item, float& ox, float& oy, float& oz, float& dx, float& dy,
                       float& dz) {
    static_assert(R == 1 || R == 2, "the work lists encode the path slot in one bit");
    const int src = static_cast<int>(item & 63u);
    const bool second = (item & 64u) != 0u;
    const float ox0 = __shfl(sl[0].p.o.x, src), oy0 = __shfl(sl[0].p.o.y, src), oz0 = __shfl(sl[0].p.o.z, src);
    const float dx0 = __shfl(sl[0].p.du.x, src), dy0 = __shfl(sl[0].p.du.y, src), dz0 = __shfl(sl[0].p.du.z, src);
    if constexpr (R == 1) {
        ox = ox0; oy = oy0; oz = oz0;
        dx = dx0; dy = dy0; dz = dz0;
        return;
    }
    const float ox1 = __shfl(sl[R - 1].p.o.x, src), oy1 = __shfl(sl[R - 1].p.o.y, src), oz1 = __shfl(sl[R - 1].p.o.z, src);
    const float dx1 = __shfl(sl[R - 1].p.du.x, src), dy1 = __shfl(sl[R - 1].p.du.y, src), dz1 = __shfl(sl[R - 1].p.du.z, src);
    ox = second ? ox1 : ox0;
    oy = second ? oy1 : oy0;
    oz = second ? oz1 : oz0;
    dx = second ? dx1 : dx0;
    dy = second ? dy1 : dy0;
    dz = second ? dz1 : dz0;
}

// Phase 2 is where rays diverge: a ray reaches 1.4 clusters on average, the unluckiest of a wave's 64
// several times that, so a loop "each lane walks its own clusters" runs at 25 % utilisation.  Instead
// the wave pools its (ray, cluster) pairs in an LDS list — a prefix sum over the lanes' candidate counts
// gives every lane the place of its items — and works through the list 64 items at a time, one
// cluster per lane: the lane fetches the ray of the item (ds_bpermute), tests the 16 members and folds
// the hit into the ray's result with an LDS atomic minimum on the packed key.  The minimum is
// order-independent, so the result is the one the per-lane walk gives.
// An item is lane | slot << 6 | (cluster - cluster_base) << 7.
template <int R>
DI void consume_items(const float4* slots, const SlotIndex* idx_map, const PathArgs& a, const uint16_t* items,
                      uint32_t total, uint32_t cluster_base, const Slot (&sl)[R], unsigned long long* results,
                      uint32_t& n_tests, uint32_t& dbg_slow_trips) {
    const uint32_t lane = threadIdx.x & 63u;
    // a wave's LDS operations are performed in order: the list is complete for the reads below
    for (uint32_t k0 = 0; k0 < total; k0 += 64u) {
        DBG_ADD(dbg_slow_trips, lane == 0u ? 1u : 0u);
        const bool valid = k0 + lane < total;
        const uint32_t item = valid ? items[k0 + lane] : 0u;
        float ox, oy, oz, dx, dy, dz;
        fetch_item_ray(sl, item, ox, oy, oz, dx, dy, dz);
        if (valid) {
            unsigned long long k2 = ~0ull;
            examine_cluster(slots, idx_map, a.n_large_slots + (cluster_base + (item >> 7)) * kClusterStride, lane, ox,
                            oy, oz, dx, dy, dz, k2);
            n_tests += kClusterSize;
            if (k2 != ~0ull) atomicMin(&results[(item & 127u)], k2);  // ds_min_u64; slot * 64 + lane
        }
    }
}

// every cluster of cm (bit 31 = cluster g0), lane by lane: the fallback when a work list overflows
DI void walk_clusters(const float4* slots, const SlotIndex* idx_map, const PathArgs& a, uint32_t cm, uint32_t g0,
                      const Path& p, unsigned long long& key, uint32_t& n_tests) {
    const uint32_t lane = threadIdx.x & 63u;
    while (cm) {
        const uint32_t bit = static_cast<uint32_t>(__builtin_clz(cm));
        cm &= ~(0x80000000u >> bit);
        examine_cluster(slots, idx_map, a.n_large_slots + (g0 + bit) * kClusterStride, lane, p.o.x, p.o.y, p.o.z,
                        p.du.x, p.du.y, p.du.z, key);
        n_tests += kClusterSize;
    }
}

// SUPER: the instantiation can meet super-clusters (a.n_super != 0).  Small scenes -- those whose shading records sit
// in LDS, at most ~510 spheres -- never have them (more than kSuperFrom clusters = more than 512 small spheres): their kernel is compiled
// without that level, which is a third of this function and would otherwise weigh on its register allocation.
// FLAT: the boxes are tested without their flat axis (slab_gap_flat; `bounds` then holds one float4 per box).
template <int R, bool SUPER, bool FLAT>
DI void trace_clustered(const float4* slots, const SlotIndex* idx_map, const float4* bounds, const PathArgs& a,
                        uint16_t* items, unsigned long long* results,
                        Slot (&sl)[R], float (&best)[R], int (&best_i)[R], uint32_t (&best_o)[R],
                        uint32_t& n_tests, uint32_t& dbg_slow_trips, uint32_t& dbg_cands,
                        unsigned long long& dbg_t_slow) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long key[R];
    float ix[R], iy[R], iz[R], ax[R], ay[R], az[R];
    bool outside[R];  // ray origin beyond the range the boxes were inflated for: take every cluster
    [[maybe_unused]] float far_add[R];  // (SUPER) ... of the top level, and the cluster boxes enlarged by this much (0: within the range)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        key[r] = ~0ull;
        results[r * 64 + lane] = ~0ull;
        const Path& p = sl[r].p;
        ix[r] = slab_rcp(p.du.x);
        iy[r] = slab_rcp(p.du.y);
        iz[r] = slab_rcp(p.du.z);
        ax[r] = -p.o.x * ix[r];
        ay[r] = -p.o.y * iy[r];
        az[r] = -p.o.z * iz[r];
        const float qx = p.o.x - a.ccenter[0], qy = p.o.y - a.ccenter[1], qz = p.o.z - a.ccenter[2];
        outside[r] = !(fma_(qz, qz, fma_(qy, qy, qx * qx)) <= a.crmax2);
        if constexpr (SUPER) {
            // Large scenes: a ray that starts outside the range takes every SUPER-cluster (their lock-step test says nothing about
            // it), but in the expansion stage below -- one lane per (ray, super-cluster) pair anyway -- its cluster boxes are tested
            // enlarged by what a ray from that far needs (far_box_margin): C5's horizon starts a few rays in a thousand out there,
            // and each of them used to take all 256 clusters, 4096 member tests.
            far_add[r] = outside[r] ? far_box_margin(a, p.o.x, p.o.y, p.o.z) : 0.0f;
        }
        if (FLAT) {  // (ix, iz, ax, az become the two box axes; iy, ay the ray's interval in the common slab)
            const FlatRay f = flat_ray(a, ix[r], iy[r], iz[r], ax[r], ay[r], az[r], SUPER ? far_add[r] : 0.0f);
            ix[r] = f.ia;
            iz[r] = f.ib;
            ax[r] = f.oa;
            az[r] = f.ob;
            iy[r] = f.tn_f;
            ay[r] = f.tf_f;
        }
    }
    // one box against the ray of slot r: sign bit set = not reached
    auto gap_of = [&](const float4& mid, const float4& half, int r) {
        return FLAT ? slab_gap_flat(mid, ix[r], iz[r], ax[r], az[r], iy[r], ay[r])
                    : slab_gap(mid, half, ix[r], iy[r], iz[r], ax[r], ay[r], az[r]);
    };
    // ---- phase 0: the large spheres, every ray, exact ----
    // A few of them (the cover scene has four: the ground and the three big balls): straight through the exact test,
    // in lock-step, no candidate stage.  Nearly every ray is a candidate for the ground anyway (its line meets the huge
    // sphere somewhere), so the candidate loop below ran two or three divergent trips of the same arithmetic per slot.
    // A discriminant that is negative, -0 or NaN gives no hit, as in examine_keyed: the sign test is explicit, and the
    // square root of a NaN fails both comparisons.
    const uint32_t n_lock = a.n_large <= kLargeLockstep ? a.n_large : 0u;
    if (n_lock != 0u) {
        const uint32_t jn4 = (n_lock + 3u) & ~3u;  // (the list is padded with slots no ray can hit: r^2 = -inf)
        for (uint32_t j = 0; j < jn4; j += 4u) {
            float4 s4[4];
            uint32_t lo4[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                s4[u] = slots[j + u];                        // wave-uniform addresses: LDS broadcast
                lo4[u] = (static_cast<uint32_t>(idx_map[j + u]) << 16) | (j + u);   // low word of the key: original index, slot
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const float4 s = s4[u];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const Path& p = sl[r].p;
                    const float ocx = p.o.x - s.x, ocy = p.o.y - s.y, ocz = p.o.z - s.z;
                    const float hb = fma_(ocz, p.du.z, fma_(ocy, p.du.y, ocx * p.du.x));
                    const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
                    const float disc = fma_(hb, hb, -cc);
                    const float sq = psqrt(disc);
                    float root = -hb - sq;
                    root = root > kTMin ? root : -hb + sq;
                    const bool hit = static_cast<int32_t>(__float_as_uint(disc)) >= 0 && root > kTMin;
                    const unsigned long long k2 =
                        hit ? (static_cast<unsigned long long>(__float_as_uint(root)) << 32) | lo4[u] : ~0ull;
                    key[r] = k2 < key[r] ? k2 : key[r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (sl[r].active) n_tests += n_lock;
    }
    for (uint32_t base = 0; base < a.n_large - n_lock; base += 32u) {
        const uint32_t jn = a.n_large - base < 32u ? a.n_large - base : 32u;
        // four spheres per trip, their reads issued together; the list is padded (to a multiple of
        // kClusterSize) with slots no ray can hit, whose discriminant is -inf or NaN
        const uint32_t jn4 = (jn + 3u) & ~3u;
        uint32_t miss[R];
#pragma unroll
        for (int r = 0; r < R; ++r) miss[r] = 0u;
        for (uint32_t j = 0; j < jn4; j += 4u) {
            float4 s4[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) s4[u] = slots[base + j + u];  // wave-uniform address: LDS broadcast
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const float4 s = s4[u];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const Path& p = sl[r].p;
                    const float ocx = p.o.x - s.x, ocy = p.o.y - s.y, ocz = p.o.z - s.z;
                    const float hb = fma_(ocz, p.du.z, fma_(ocy, p.du.y, ocx * p.du.x));
                    const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
                    const float disc = fma_(hb, hb, -cc);
                    // (a NaN discriminant may carry either sign: examine_keyed looks again)
                    miss[r] = __builtin_amdgcn_alignbit(miss[r], __float_as_uint(disc), 31);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t cand = sl[r].active ? ~miss[r] << (32u - jn4) : 0u;  // first sphere of the word at bit 31
            if (sl[r].active) n_tests += jn;
            const Path& p = sl[r].p;
            while (cand) {
                const uint32_t bit = static_cast<uint32_t>(__builtin_clz(cand));
                cand &= ~(0x80000000u >> bit);
                examine_keyed(slots, idx_map, base + bit, p.o.x, p.o.y, p.o.z, p.du.x, p.du.y, p.du.z, key[r]);
            }
        }
    }
#ifndef RTIOW_CLAMP_TO_LARGE_HIT
#define RTIOW_CLAMP_TO_LARGE_HIT 1  // (-DRTIOW_CLAMP_TO_LARGE_HIT=0: A/B only)
#endif
    if constexpr (FLAT && RTIOW_CLAMP_TO_LARGE_HIT != 0) {
        // Round 5: a ray that has hit a large sphere at t_L -- the ground, for every ray that points down -- needs no box beyond t_L: a
        // member that is hit closer (or as close: the tie goes by index) has its computed hit point inside its box -- that is what the
        // boxes are inflated for (rtiow_clusters.cpp) -- so the ray enters that box no later than t_L.  The ray's far end in the common
        // slab of a flat scene is one value per ray (FlatRay::tf_f): clamping it costs nothing per box.
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float t_large = key[r] != ~0ull ? __uint_as_float(static_cast<uint32_t>(key[r] >> 32)) * 1.000001f : __builtin_inff();
            ay[r] = __builtin_fminf(ay[r], t_large);
        }
    }
    if (!SUPER || a.n_super == 0u) {
        // ---- phases 1 and 2, 32 clusters at a time ----
        for (uint32_t g0 = 0; g0 < a.n_clusters; g0 += 32u) {
            uint32_t miss[R];
#pragma unroll
            for (int r = 0; r < R; ++r) miss[r] = 0u;
            const uint32_t jn = a.n_clusters - g0 < 32u ? a.n_clusters - g0 : 32u;  // n_clusters is a multiple of 8
            // four boxes per trip, their eight ds_read_b128 issued together (one at a time the loop waited out
            // the LDS latency for every box; and the compiler read the six floats it needs as two ds_read_b96,
            // which occupy the LDS twice as long as two ds_read_b128 -- hence keep_b128)
            for (uint32_t j = 0; j < jn; j += 4u) {
                float4 mid[4], half[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    if (FLAT) {
                        mid[u] = half[u] = bounds[g0 + j + u];  // LDS broadcast: one read per box
                    } else {
                        mid[u] = bounds[2u * (g0 + j + u)];
                        half[u] = bounds[2u * (g0 + j + u) + 1u];  // LDS broadcast
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        miss[r] = __builtin_amdgcn_alignbit(miss[r], __float_as_uint(gap_of(mid[u], half[u], r)), 31);
                    if (!FLAT) {
                        keep_b128(mid[u]);   // (after the use: the asm waits for the read)
                        keep_b128(half[u]);
                    }
                }
            }
            [[maybe_unused]] const unsigned long long ts0 = DBG_STAMP();
            uint32_t cm[R];
            uint32_t packed = 0u;  // candidate clusters of the lane, both slots
#pragma unroll
            for (int r = 0; r < R; ++r) {
                uint32_t far_cm = ~0u;  // the clusters a ray from outside the boxes' range takes
#if RTIOW_FAR_LOCKSTEP
                // One wave-wide test per far ray instead of "every cluster": the ray goes to scalars (v_readlane), lane b tests box
                // g0 + b enlarged by the ray's margin (far_box_margin), one ballot is the ray's candidate mask.  On the cover frame one
                // scattered ray in a hundred starts out there -- the ground toward the horizon -- and its 32 items were a seventh of
                // all the work of the member stage.
                unsigned long long fm = __ballot(sl[r].active && outside[r]);
                while (fm != 0ull) {
                    const int l = __builtin_ctzll(fm);
                    fm &= fm - 1ull;
                    const Path& p = sl[r].p;
                    auto of_lane = [&](float v) {
                        return __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(v)), l)));
                    };
                    const float ox = of_lane(p.o.x), oy = of_lane(p.o.y), oz = of_lane(p.o.z);
                    const float dx = of_lane(p.du.x), dy = of_lane(p.du.y), dz = of_lane(p.du.z);
                    const uint32_t b = lane & 31u;
                    const bool reach = b < jn && reaches_enlarged<FLAT>(bounds, a, g0 + b, ox, oy, oz, dx, dy, dz, far_box_margin(a, ox, oy, oz));
                    const uint32_t m32 = static_cast<uint32_t>(__ballot(reach));  // bit b: cluster g0 + b (lanes 32-63 repeat 0-31)
                    if (lane == static_cast<uint32_t>(l)) far_cm = __builtin_bitreverse32(m32);  // first cluster at bit 31, as below
                }
                cm[r] = sl[r].active ? (outside[r] ? far_cm : (~miss[r] << (32u - jn))) : 0u;
#else
                cm[r] = sl[r].active ? (outside[r] ? far_cm : ~miss[r]) << (32u - jn) : 0u;  // first cluster at bit 31
#endif
                if (sl[r].active) n_tests += jn;
                packed += static_cast<uint32_t>(__builtin_popcount(cm[r]));
                DBG_ADD(dbg_cands, __builtin_popcount(cm[r]));
            }
            const uint32_t incl = wave_inclusive_sum(packed);  // over the lanes
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            if (total == 0u) continue;
            if (total > kItemCap) {  // (only when many rays start outside the boxes' range)
#pragma unroll
                for (int r = 0; r < R; ++r) walk_clusters(slots, idx_map, a, cm[r], g0, sl[r].p, key[r], n_tests);
                continue;
            }
            {   // my items, at my place in the list
                uint32_t pos = incl - packed;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    uint32_t m = cm[r];
                    while (m) {
                        const uint32_t bit = static_cast<uint32_t>(__builtin_clz(m));
                        m &= ~(0x80000000u >> bit);
                        items[pos++] = static_cast<uint16_t>(lane | (static_cast<uint32_t>(r) << 6) | (bit << 7));
                    }
                }
            }
            consume_items(slots, idx_map, a, items, total, g0, sl, results, n_tests, dbg_slow_trips);
            DBG_ADD(dbg_t_slow, DBG_STAMP() - ts0);
        }
    } else {
        // ---- large scenes: super-cluster boxes in lock-step (32 at a time: 256 clusters), then the
        // cluster boxes of the (ray, super-cluster) pairs that pass, one pair per lane, then the members ----
        static_assert(64u * kSuperSize <= kItemCap, "one round of (ray, super-cluster) items must fit the cluster list");
        static_assert(32u * kSuperSize <= 512u, "cluster-in-group index must fit the 9 bits above lane and slot");
        const float4* sbounds = bounds + (FLAT ? 1u : 2u) * a.n_clusters;
        uint16_t* sitems = items + kItemCap;
        for (uint32_t s0 = 0; s0 < a.n_super; s0 += 32u) {
            uint32_t miss[R];
#pragma unroll
            for (int r = 0; r < R; ++r) miss[r] = 0u;
            const uint32_t jn = a.n_super - s0 < 32u ? a.n_super - s0 : 32u;
            uint32_t j = 0;
            for (; j + 4u <= jn; j += 4u) {  // four boxes per trip, as the cluster boxes above
                float4 mid[4], half[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    if (FLAT) {
                        mid[u] = half[u] = sbounds[s0 + j + u];  // LDS broadcast: one read per box
                    } else {
                        mid[u] = sbounds[2u * (s0 + j + u)];
                        half[u] = sbounds[2u * (s0 + j + u) + 1u];  // LDS broadcast
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        miss[r] = __builtin_amdgcn_alignbit(miss[r], __float_as_uint(gap_of(mid[u], half[u], r)), 31);
                    if (!FLAT) {
                        keep_b128(mid[u]);
                        keep_b128(half[u]);
                    }
                }
            }
            for (; j < jn; ++j) {
                const float4 mid = FLAT ? sbounds[s0 + j] : sbounds[2u * (s0 + j)];  // LDS broadcast
                const float4 half = FLAT ? mid : sbounds[2u * (s0 + j) + 1u];
#pragma unroll
                for (int r = 0; r < R; ++r) miss[r] = __builtin_amdgcn_alignbit(miss[r], __float_as_uint(gap_of(mid, half, r)), 31);
            }
            [[maybe_unused]] const unsigned long long ts0 = DBG_STAMP();
            uint32_t sm[R];
            uint32_t packed = 0u;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                sm[r] = sl[r].active ? (outside[r] ? ~0u : ~miss[r]) << (32u - jn) : 0u;  // first super-cluster at bit 31
                if (sl[r].active) n_tests += jn;
                packed += static_cast<uint32_t>(__builtin_popcount(sm[r]));
            }
            const uint32_t incl_all = wave_inclusive_sum(packed);
            const uint32_t total_all = __builtin_amdgcn_readlane(incl_all, 63);
            if (total_all == 0u) continue;
            // A list that would overflow -- many rays that start outside the boxes' range: 32 items each -- is worked off sixteen
            // lanes of one slot at a time (16 x 32 items fit): every ray still goes through the expansion stage, where a far ray's
            // boxes are tested enlarged.  (Rounds 1-3 walked every cluster of every candidate here, lane by lane.)
            const uint32_t n_batches = total_all > kItemCap ? 4u * static_cast<uint32_t>(R) : 1u;
            static_assert(16u * 32u <= kItemCap, "sixteen lanes' super-cluster items fit the list");
            for (uint32_t bt = 0; bt < n_batches; ++bt) {
            uint32_t part[R];
            uint32_t packed_b = 0u;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                part[r] = n_batches == 1u || (static_cast<uint32_t>(r) == (bt >> 2) && (lane >> 4) == (bt & 3u)) ? sm[r] : 0u;
                packed_b += static_cast<uint32_t>(__builtin_popcount(part[r]));
            }
            const uint32_t incl = n_batches == 1u ? incl_all : wave_inclusive_sum(packed_b);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            if (total == 0u) continue;
            {
                uint32_t pos = incl - packed_b;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    uint32_t m = part[r];
                    while (m) {
                        const uint32_t bit = static_cast<uint32_t>(__builtin_clz(m));
                        m &= ~(0x80000000u >> bit);
                        sitems[pos++] = static_cast<uint16_t>(lane | (static_cast<uint32_t>(r) << 6) | (bit << 7));
                    }
                }
            }
            uint32_t pending = 0u;  // (ray, cluster) items in `items`
            for (uint32_t k0 = 0; k0 < total; k0 += 64u) {
                const bool valid = k0 + lane < total;
                const uint32_t item = valid ? sitems[k0 + lane] : 0u;
                uint32_t hit8 = 0u;  // bit k: cluster k of the super-cluster
                if constexpr (FLAT) {
                    // the item's flat ray straight from the registers of the lane that owns the ray (the same twelve ds_bpermute as
                    // its origin and direction, without working the reciprocals, the range test and the flat interval out again:
                    // cover scenes of 785 / 2304 / 4099 spheres -2.9 / -2.0 / -1.8 %)
                    const int src = static_cast<int>(item & 63u);
                    const bool second = R > 1 && (item & 64u) != 0u;
                    float fr[7];
                    const float* own[7] = {ix, iz, ax, az, iy, ay, far_add};
#pragma unroll
                    for (int v = 0; v < 7; ++v) {
                        const float v0 = __shfl(own[v][0], src), v1 = __shfl(own[v][R - 1], src);
                        fr[v] = second ? v1 : v0;
                    }
                    if (valid) {
                        const uint32_t first = (s0 + (item >> 7)) * kSuperSize;
#pragma unroll
                        for (uint32_t j = 0; j < kSuperSize; ++j) {
                            const uint32_t k = (j ^ lane) & (kSuperSize - 1u);  // lane-permuted (an XOR: see examine_cluster): spreads the LDS banks
                            float4 b = bounds[first + k];
                            b.z += fr[6];  // (0 for a ray within the range: the box as stored)
                            b.w += fr[6];
                            const bool reach = !__builtin_signbit(slab_gap_flat(b, fr[0], fr[1], fr[2], fr[3], fr[4], fr[5]));
                            hit8 |= (reach ? 1u : 0u) << k;
                        }
                        n_tests += kSuperSize;
                        DBG_ADD(dbg_cands, __builtin_popcount(hit8));
                    }
                }
                if constexpr (!FLAT) {  // the same with whole boxes: reciprocal direction and -o/d of the item's ray from its owner
                    const int src = static_cast<int>(item & 63u);
                    const bool second = R > 1 && (item & 64u) != 0u;
                    float fr[7];
                    const float* own[7] = {ix, iy, iz, ax, ay, az, far_add};
#pragma unroll
                    for (int v = 0; v < 7; ++v) {
                        const float v0 = __shfl(own[v][0], src), v1 = __shfl(own[v][R - 1], src);
                        fr[v] = second ? v1 : v0;
                    }
                    if (valid) {
                        const uint32_t first = (s0 + (item >> 7)) * kSuperSize;
#pragma unroll
                        for (uint32_t j = 0; j < kSuperSize; ++j) {
                            const uint32_t k = (j ^ lane) & (kSuperSize - 1u);  // lane-permuted (an XOR: see examine_cluster): spreads the LDS banks
                            const float4 mid = bounds[2u * (first + k)];
                            float4 half = bounds[2u * (first + k) + 1u];
                            half.x += fr[6];  // (0 for a ray within the range: the box as stored)
                            half.y += fr[6];
                            half.z += fr[6];
                            const bool reach = !__builtin_signbit(slab_gap(mid, half, fr[0], fr[1], fr[2], fr[3], fr[4], fr[5]));
                            hit8 |= (reach ? 1u : 0u) << k;
                        }
                        n_tests += kSuperSize;
                        DBG_ADD(dbg_cands, __builtin_popcount(hit8));
                    }
                }
                const uint32_t cnt = static_cast<uint32_t>(__builtin_popcount(hit8));
                const uint32_t incl2 = wave_inclusive_sum(cnt);
                const uint32_t tot2 = __builtin_amdgcn_readlane(incl2, 63);  // <= 64 * 8 = kItemCap
                if (pending + tot2 > kItemCap) {
                    consume_items(slots, idx_map, a, items, pending, s0 * kSuperSize, sl, results, n_tests, dbg_slow_trips);
                    pending = 0u;
                }
                uint32_t pos = pending + incl2 - cnt;
                while (hit8) {
                    const uint32_t k = static_cast<uint32_t>(__builtin_ctz(hit8));
                    hit8 &= hit8 - 1u;
                    items[pos++] = static_cast<uint16_t>((item & 127u) | (((item >> 7) * kSuperSize + k) << 7));
                }
                pending += tot2;
            }
            consume_items(slots, idx_map, a, items, pending, s0 * kSuperSize, sl, results, n_tests, dbg_slow_trips);
            }  // (batches)
            DBG_ADD(dbg_t_slow, DBG_STAMP() - ts0);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long pooled = results[r * 64 + lane];
        const unsigned long long k = pooled < key[r] ? pooled : key[r];
        const bool hit = k != ~0ull;
        best[r] = hit ? __uint_as_float(static_cast<uint32_t>(k >> 32)) : __builtin_inff();
        best_i[r] = hit ? static_cast<int>(k & 0xFFFFu) : -1;
        best_o[r] = static_cast<uint32_t>(k >> 16) & 0xFFFFu;
    }
}

// A 64-bit constant (both words `word`) made where it is stored.  Left to itself the compiler makes the register pair in the kernel's first
// block and keeps it to the last; the COMPACT variant, at its 128 registers, spilled two such pairs (0 for a fresh accumulator entry, ~0 for
// fresh result keys) to scratch and read them back inside the loops: its only scratch traffic.  Two v_mov at the store instead.
DI unsigned long long fresh64(uint32_t word) {
    uint32_t lo = word, hi = word;
    asm volatile("" : "+v"(lo), "+v"(hi));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

DI uint32_t lane_rank(unsigned long long mask) {  // number of set bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// Path-parallel closest hit for a wave with few live paths (at most kSparseParMax): the end of a frame
// or tile, where what counts is the latency of one iteration of the longest paths.  round 1's cluster-parallel trace
// took one path at a time (about 1 us each); here the wave's P live paths are worked on together.  Their
// rays go to LDS and every path gets a group of 64 / P lanes, which strides over the large spheres (exact
// test) and the top-level boxes (the cluster boxes, or the super-cluster boxes of a large scene) with the
// path's ray in registers; the boxes a ray reaches are appended to a work list (ballot + mbcnt give the places);
// super-cluster items are expanded, eight lanes per item, into cluster items; cluster items are consumed
// sixteen lanes per item, one member per lane.  Every hit is folded into its path's key with ds_min_u64:
// same tests, same key, same minimum as trace_clustered.  A list that fills up is drained on the spot.
constexpr uint32_t kSparseParMax = 32;  // paths; their keys and rays fill the wave's 1 KiB result area
static_assert(kSparseParMax * 8u + kSparseParMax * 24u <= kWaveResultBytes, "keys + rays must fit the result area");

struct SparseRay {
    float ox, oy, oz, dx, dy, dz;
};
DI SparseRay load_sparse_ray(const float* rays, uint32_t p) {
    const float2* r = reinterpret_cast<const float2*>(rays + p * 6u);  // 24-byte records: three ds_read_b64
    const float2 a = r[0], b = r[1], c = r[2];
    return SparseRay{a.x, a.y, b.x, b.y, c.x, c.y};
}

// consumes `count` (path, cluster) items: sixteen lanes per item, one member each
DI void sparse_members(const float4* slots, const SlotIndex* idx_map, const PathArgs& a, const uint16_t* items,
                       uint32_t count, const float* rays, unsigned long long* keys, uint32_t& n_tests) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = count * kClusterSize;
    for (uint32_t q0 = 0; q0 < total; q0 += 64u) {
        const uint32_t q = q0 + lane;
        if (q < total) {
            const uint32_t item = items[q / kClusterSize];
            const uint32_t p = item & (kSparseParMax - 1u);
            const SparseRay ray = load_sparse_ray(rays, p);
            unsigned long long k2 = ~0ull;
            examine_keyed(slots, idx_map, a.n_large_slots + (item >> 5) * kClusterStride + (q & (kClusterSize - 1u)),
                          ray.ox, ray.oy, ray.oz, ray.dx, ray.dy, ray.dz, k2);
            ++n_tests;
            if (k2 != ~0ull) atomicMin(&keys[p], k2);
        }
    }
}

template <int R, bool SUPER, bool FLAT>
DI void trace_sparse_parallel(const float4* slots, const SlotIndex* idx_map, const float4* bounds, const PathArgs& a,
                              uint16_t* items, unsigned long long* results, Slot (&sl)[R], float (&best)[R],
                              int (&best_i)[R], uint32_t (&best_o)[R], uint32_t& n_tests) {
    static_assert(kSparseParMax == 32u, "items carry the path in five bits");
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long* keys = results;
    float* rays = reinterpret_cast<float*>(results + kSparseParMax);
    uint32_t pidx[R];
    uint32_t n_live = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long m = __ballot(sl[r].active);
        pidx[r] = n_live + lane_rank(m);
        n_live += static_cast<uint32_t>(__popcll(m));
        if (sl[r].active) {
            const Path& p = sl[r].p;
            float2* dst = reinterpret_cast<float2*>(rays + pidx[r] * 6u);
            dst[0] = make_float2(p.o.x, p.o.y);
            dst[1] = make_float2(p.o.z, p.du.x);
            dst[2] = make_float2(p.du.y, p.du.z);
        }
    }
    if (lane < n_live) keys[lane] = fresh64(~0u);
    // (a wave's LDS operations are performed in order: rays and keys are in place for what follows)
    const bool two_level = SUPER && a.n_super != 0u;
    const uint32_t top = two_level ? a.n_clusters : 0u;  // number of the first box of the top level
    const uint32_t n_top = two_level ? a.n_super : a.n_clusters;
    uint16_t* top_items = two_level ? items + kItemCap : items;
    uint32_t pending = 0u;   // items in top_items
    uint32_t pending2 = 0u;  // two_level: cluster items in `items`

    // expands the super-cluster items into cluster items (eight lanes per item), members as the list fills
    auto expand_supers = [&]() {
        const uint32_t total = pending * kSuperSize;
        for (uint32_t q0 = 0; q0 < total; q0 += 64u) {
            if (pending2 + 64u > kItemCap) {
                sparse_members(slots, idx_map, a, items, pending2, rays, keys, n_tests);
                pending2 = 0u;
            }
            const uint32_t q = q0 + lane;
            bool reach = false;
            uint32_t out_item = 0u;
            if (q < total) {
                const uint32_t item = top_items[q / kSuperSize];
                const uint32_t p = item & (kSparseParMax - 1u);
                const uint32_t c = (item >> 5) * kSuperSize + (q & (kSuperSize - 1u));
                const SparseRay ray = load_sparse_ray(rays, p);
                const float ix = slab_rcp(ray.dx), iy = slab_rcp(ray.dy), iz = slab_rcp(ray.dz);
                const float qx = ray.ox - a.ccenter[0], qy = ray.oy - a.ccenter[1], qz = ray.oz - a.ccenter[2];
                // (a path that starts outside the boxes' range: the box enlarged by its margin, not every box -- far_box_margin)
                const float far_add = !(fma_(qz, qz, fma_(qy, qy, qx * qx)) <= a.crmax2) ? far_box_margin(a, ray.ox, ray.oy, ray.oz) : 0.0f;
                float4 mid, half;
                load_box<FLAT>(bounds, a, c, mid, half);
                half.x += far_add;
                half.y += far_add;
                half.z += far_add;
                reach = !__builtin_signbit(slab_gap(mid, half, ix, iy, iz, -ray.ox * ix, -ray.oy * iy, -ray.oz * iz));
                ++n_tests;
                out_item = p | (c << 5);
            }
            const unsigned long long m = __ballot(reach);
            if (reach) items[pending2 + lane_rank(m)] = static_cast<uint16_t>(out_item);
            pending2 += static_cast<uint32_t>(__popcll(m));
        }
        pending = 0u;
    };
    auto drain = [&]() {
        if (two_level) {
            expand_supers();
        } else {
            sparse_members(slots, idx_map, a, items, pending, rays, keys, n_tests);
            pending = 0u;
        }
    };

    // Every path gets a group of 64 / P' consecutive lanes (P' = P rounded up to a power of two): a lane keeps
    // its path -- ray, reciprocal direction and all -- and strides over the objects of the top level.
    uint32_t shift = 6u;  // log2 of the group size
    while ((n_live << shift) > 64u) --shift;
    const uint32_t gsz = 1u << shift;
    const uint32_t p = lane >> shift, j0 = lane & (gsz - 1u);
    const bool mine = p < n_live;
    const SparseRay ray = load_sparse_ray(rays, mine ? p : 0u);
    const float ix = slab_rcp(ray.dx), iy = slab_rcp(ray.dy), iz = slab_rcp(ray.dz);
    const float ax = -ray.ox * ix, ay = -ray.oy * iy, az = -ray.oz * iz;
    const float qx = ray.ox - a.ccenter[0], qy = ray.oy - a.ccenter[1], qz = ray.oz - a.ccenter[2];
    // a path that starts outside the range the boxes were inflated for tests them enlarged by its own margin (far_box_margin: 0 within
    // the range); it used to take every box -- one such path, a ground bounce near the horizon, made a sparse iteration three times as long
    const float far_add = !(fma_(qz, qz, fma_(qy, qy, qx * qx)) <= a.crmax2) ? far_box_margin(a, ray.ox, ray.oy, ray.oz) : 0.0f;
    // the large spheres, exactly
    for (uint32_t base = 0; base < a.n_large; base += gsz) {
        const uint32_t j = base + j0;
        if (mine && j < a.n_large) {
            unsigned long long k2 = ~0ull;
            examine_keyed(slots, idx_map, j, ray.ox, ray.oy, ray.oz, ray.dx, ray.dy, ray.dz, k2);
            ++n_tests;
            if (k2 != ~0ull) atomicMin(&keys[p], k2);
        }
    }
    // the top-level boxes
    for (uint32_t base = 0; base < n_top; base += gsz) {
        if (pending + 64u > kItemCap) drain();
        const uint32_t b = base + j0;
        bool reach = false;
        if (mine && b < n_top) {
            float4 mid, half;
            load_box<FLAT>(bounds, a, top + b, mid, half);
            half.x += far_add;
            half.y += far_add;
            half.z += far_add;
            reach = !__builtin_signbit(slab_gap(mid, half, ix, iy, iz, ax, ay, az));
            ++n_tests;
        }
        const unsigned long long m = __ballot(reach);
        if (reach) top_items[pending + lane_rank(m)] = static_cast<uint16_t>(p | (b << 5));
        pending += static_cast<uint32_t>(__popcll(m));
    }
    drain();
    if (two_level) sparse_members(slots, idx_map, a, items, pending2, rays, keys, n_tests);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long k = sl[r].active ? keys[pidx[r]] : ~0ull;
        const bool hit = k != ~0ull;
        best[r] = hit ? __uint_as_float(static_cast<uint32_t>(k >> 32)) : __builtin_inff();
        best_i[r] = hit ? static_cast<int>(k & 0xFFFFu) : -1;
        best_o[r] = static_cast<uint32_t>(k >> 16) & 0xFFFFu;
    }
}

// ---- The same for at most kSparseBatchMax paths, with the LDS round trips taken together --------------------------------------
// trace_sparse_parallel takes a path through the boxes one at a time -- read, test, ballot, write the item, and only then
// the next read -- and through the work list 64 lanes at a time: with P paths on 32 boxes that is P / 2 + P / 3 dependent
// LDS round trips of a few hundred cycles each, and at the end of a frame, where a wave is alone on its SIMD, those are what
// an iteration costs (tools/sparse_latency.py: 2.0 us for 3 paths, 3.7 for a dozen, 6.3 for 32 in a lone wave).  Here a lane takes FOUR boxes per trip --
// their reads issued together, one prefix sum over the lanes' counts places all the items -- and FOUR list entries per
// trip, sphere, ray and original index of each asked for at once.  Same tests, same key, same minimum.
// One level of boxes only (the small-scene kernels, whose sparse loop is the one place that calls it).
#ifndef RTIOW_SPARSE_BATCH_MIN
#define RTIOW_SPARSE_BATCH_MIN 1
#endif
constexpr uint32_t kSparseBatchMax = 16;  // paths
// (With one to four paths a trip's four boxes and four list entries are half padding, and tools/sparse_latency.py -- every
// wave of the chip with P mirror paths -- has the batched form 6 % slower there and 8-19 % faster from six paths on; on real
// frames thresholds of 1 / 3 / 5 / 7 paths are indistinguishable -- one eighth of the cover frame 1.209 / 1.239 / 1.216 /
// 1.219 ms against 1.233 without, a 1-spp frame 0.439 / 0.445 / 0.443 / 0.452 against 0.456 -- so there is no threshold.)
constexpr uint32_t kSparseBatchMin = RTIOW_SPARSE_BATCH_MIN;

// sphere s (original index orig, list slot `slot`) against a ray: examine_keyed with its operands already loaded
DI void examine_loaded(const float4& s, uint32_t orig, uint32_t slot, const SparseRay& r, unsigned long long& key) {
    const float ocx = r.ox - s.x, ocy = r.oy - s.y, ocz = r.oz - s.z;
    const float hb = fma_(ocz, r.dz, fma_(ocy, r.dy, ocx * r.dx));
    const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
    const float disc = fma_(hb, hb, -cc);
    if (__builtin_signbit(disc) || disc != disc) return;
    const float sq = psqrt(disc);
    float root = -hb - sq;
    root = root > kTMin ? root : -hb + sq;
    if (!(root > kTMin)) return;
    const unsigned long long k2 = (static_cast<unsigned long long>(__float_as_uint(root)) << 32) | (orig << 16) | slot;
    key = k2 < key ? k2 : key;
}

// consumes `count` (path, cluster) items: sixteen lanes per item, one member each, four items per lane and trip
[[maybe_unused]] DI void sparse_members4(const float4* slots, const SlotIndex* idx_map, const PathArgs& a, const uint16_t* items,
                        uint32_t count, const float* rays, unsigned long long* keys, uint32_t& n_tests) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = count * kClusterSize;
    for (uint32_t q0 = 0; q0 < total; q0 += 256u) {
        uint32_t item[4];
        bool valid[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t q = q0 + u * 64u + lane;
            valid[u] = q < total;
            item[u] = valid[u] ? items[q / kClusterSize] : 0u;
        }
        float4 s[4];
        uint32_t orig[4], slot[4];
        SparseRay ray[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            slot[u] = a.n_large_slots + (item[u] >> 5) * kClusterStride + (lane & (kClusterSize - 1u));
            ray[u] = load_sparse_ray(rays, item[u] & (kSparseParMax - 1u));
            s[u] = slots[slot[u]];
            orig[u] = idx_map[slot[u]];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            if (valid[u]) {
                unsigned long long k2 = ~0ull;
                examine_loaded(s[u], orig[u], slot[u], ray[u], k2);
                ++n_tests;
                if (k2 != ~0ull) atomicMin(&keys[item[u] & (kSparseParMax - 1u)], k2);
            }
        }
    }
}

template <bool FLAT>
DI void trace_sparse_batched(const float4* slots, const SlotIndex* idx_map, const float4* bounds, const PathArgs& a,
                             uint16_t* items, unsigned long long* results, Slot& sl, float& best, int& best_i,
                             uint32_t& best_o, uint32_t& n_tests) {
    static_assert(kSparseBatchMax <= kSparseParMax && 4u * 64u <= kItemCap, "items carry the path in five bits; a trip's items fit the list");
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long* keys = results;
    float* rays = reinterpret_cast<float*>(results + kSparseParMax);
    const unsigned long long live = __ballot(sl.active);
    const uint32_t n_live = static_cast<uint32_t>(__popcll(live));
    const uint32_t pidx = lane_rank(live);
    if (sl.active) {
        const Path& p = sl.p;
        float2* dst = reinterpret_cast<float2*>(rays + pidx * 6u);
        dst[0] = make_float2(p.o.x, p.o.y);
        dst[1] = make_float2(p.o.z, p.du.x);
        dst[2] = make_float2(p.du.y, p.du.z);
    }
    if (lane < n_live) keys[lane] = fresh64(~0u);
    // (a wave's LDS operations are performed in order: rays and keys are in place for what follows)
    uint32_t shift = 6u;  // log2 of the lanes a path gets
    while ((n_live << shift) > 64u) --shift;
    const uint32_t gsz = 1u << shift;
    const uint32_t p = lane >> shift, j0 = lane & (gsz - 1u);
    const bool mine = p < n_live;
    const SparseRay ray = load_sparse_ray(rays, mine ? p : 0u);
    const float ix = slab_rcp(ray.dx), iy = slab_rcp(ray.dy), iz = slab_rcp(ray.dz);
    const float ax = -ray.ox * ix, ay = -ray.oy * iy, az = -ray.oz * iz;
    const float qx = ray.ox - a.ccenter[0], qy = ray.oy - a.ccenter[1], qz = ray.oz - a.ccenter[2];
    // (a path that starts outside the boxes' range: boxes and common interval enlarged by its margin, as in trace_sparse_parallel)
    const float far_add = !(fma_(qz, qz, fma_(qy, qy, qx * qx)) <= a.crmax2) ? far_box_margin(a, ray.ox, ray.oy, ray.oz) : 0.0f;
    [[maybe_unused]] FlatRay f{};
    if (FLAT) f = flat_ray(a, ix, iy, iz, ax, ay, az, far_add);
    // the large spheres, exactly
    for (uint32_t base = 0; base < a.n_large; base += gsz) {
        const uint32_t j = base + j0;
        if (mine && j < a.n_large) {
            unsigned long long k2 = ~0ull;
            examine_keyed(slots, idx_map, j, ray.ox, ray.oy, ray.oz, ray.dx, ray.dy, ray.dz, k2);
            ++n_tests;
            if (k2 != ~0ull) atomicMin(&keys[p], k2);
        }
    }
    // the cluster boxes, four per lane and trip
    uint32_t pending = 0u;
    for (uint32_t base = 0; base < a.n_clusters; base += 4u * gsz) {
        if (pending + 256u > kItemCap) {
            sparse_members4(slots, idx_map, a, items, pending, rays, keys, n_tests);
            pending = 0u;
        }
        uint32_t b[4];
        bool valid[4];
        float4 mid[4], half[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            b[u] = base + u * gsz + j0;
            valid[u] = mine && b[u] < a.n_clusters;
            const uint32_t i = valid[u] ? b[u] : 0u;
            if (FLAT) {
                mid[u] = half[u] = bounds[i];
            } else {
                mid[u] = bounds[2u * i];
                half[u] = bounds[2u * i + 1u];
            }
        }
        uint32_t mask = 0u;
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            if (FLAT) {
                mid[u].z += far_add;
                mid[u].w += far_add;
            } else {
                half[u].x += far_add;
                half[u].y += far_add;
                half[u].z += far_add;
            }
            const float gap = FLAT ? slab_gap_flat(mid[u], f.ia, f.ib, f.oa, f.ob, f.tn_f, f.tf_f)
                                   : slab_gap(mid[u], half[u], ix, iy, iz, ax, ay, az);
            if (valid[u]) {
                ++n_tests;
                if (!__builtin_signbit(gap)) mask |= 1u << u;
            }
        }
        const uint32_t cnt = static_cast<uint32_t>(__builtin_popcount(mask));
        const uint32_t incl = wave_inclusive_sum(cnt);
        uint32_t pos = pending + incl - cnt;
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u)
            if ((mask >> u) & 1u) items[pos++] = static_cast<uint16_t>(p | (b[u] << 5));
        pending += __builtin_amdgcn_readlane(incl, 63);
    }
    sparse_members4(slots, idx_map, a, items, pending, rays, keys, n_tests);
    const unsigned long long k = sl.active ? keys[pidx] : ~0ull;
    const bool hit = k != ~0ull;
    best = hit ? __uint_as_float(static_cast<uint32_t>(k >> 32)) : __builtin_inff();
    best_i = hit ? static_cast<int>(k & 0xFFFFu) : -1;
    best_o = static_cast<uint32_t>(k >> 16) & 0xFFFFu;
}

// Moves the live paths of slot 1 into idle lanes of slot 0 through a per-wave LDS scratch (64-byte
// records).  Only for a wave with at most 32 live paths: slot 0 then has room for all of them.
// Which lane carries a path is immaterial: its pixel's accumulator is addressed by `entry`.
template <int R>
DI void compact_to_slot0(Slot (&sl)[R], uint32_t* scratch) {
    if constexpr (R == 1) return;
    const unsigned long long m1 = __ballot(sl[R - 1].active);
    if (m1 == 0ull) return;
    const uint32_t n1 = static_cast<uint32_t>(__popcll(m1));
    const unsigned long long idle0 = __ballot(!sl[0].active);
    float* recs = reinterpret_cast<float*>(scratch);
    if (sl[R - 1].active) {
        Slot& q = sl[R - 1];
        float* rec = recs + lane_rank(m1) * 16u;
        rec[0] = q.p.o.x; rec[1] = q.p.o.y; rec[2] = q.p.o.z;
        rec[3] = q.p.du.x; rec[4] = q.p.du.y; rec[5] = q.p.du.z;
        rec[6] = q.p.att.x; rec[7] = q.p.att.y; rec[8] = q.p.att.z;
        rec[9] = __uint_as_float(q.p.rng.state);
        rec[10] = __uint_as_float(q.meta);
        q.active = false;
    }
    const uint32_t k = lane_rank(idle0);
    if (!sl[0].active && k < n1) {
        Slot& q = sl[0];
        const float* rec = recs + k * 16u;
        q.p.o = mk(rec[0], rec[1], rec[2]);
        q.p.du = mk(rec[3], rec[4], rec[5]);
        q.p.att = mk(rec[6], rec[7], rec[8]);
        q.p.rng = Pcg(__float_as_uint(rec[9]));
        q.meta = __float_as_uint(rec[10]);
        q.active = true;
    }
}

// ============================================================================
// The primary pass (clustered list): camera rays are traced where they are made
// ============================================================================
// A third of all segments are camera rays, and the camera rays a wave starts together are consecutive samples of
// ONE pixel (or of a few neighbours): unlike the scattered rays in the slots they all pass through the same small
// cone.  So the wave works out ONCE which clusters that cone can reach -- lane b takes box b, one ballot gives a
// wave-uniform mask -- and every new ray then tests the members of those clusters, and those only, in lock-step
// with LDS broadcast reads: no box test per ray, no work list, no divergence.  The rays are shaded on the spot;
// the paths that go on (at depth 1) are handed to idle slots through LDS, so that the general trace -- 32 box tests
// per ray, work lists, per-lane member walks -- only ever sees scattered rays and always runs on full slots.  A sky
// pixel never reaches the slots at all.
//
// The cone test is only a cull, like the per-ray slab test: it must never drop a cluster that holds a sphere the
// exact test would accept for some ray of the span.  Rays of a span of pixels i_lo..i_hi of one row (camera_path):
// origin O + off with |off| <= rho_L (the lens), through the focal-plane point T of its pixel, |T - Tc| <= rho_T
// for the span's centre Tc.  At parameter l (1 at the focal plane) the ray's point differs from the axis point
// O + l (Tc - O) by (1 - l) off + l (T - Tc), at most rho_L + l (rho_L + rho_T) = rho_L + kappa s in terms of the
// distance s along the axis.  A ray point inside a box (centre c, half extent h) therefore has its axis point inside
// the box inflated by rad = rho_L + kappa s_far, s_far = (max(e.Dn, 0) + |h|_1 + rho_L) / (1 - kappa) bounding s:
// the AXIS is put through the ordinary slab test against that inflated box.  rho_T, rho_L and rad carry relative
// margins of 2^-6 .. 2^-8 and an absolute one of 2^-16 of the scene's coordinate range, three orders of magnitude
// above the binary32 rounding of camera_path and of this test.  The boxes themselves already contain every sphere
// inflated for the exact test's own rounding (rtiow_clusters.cpp), for origins within their range: launch_path
// switches the cull off (every cluster, every ray) for a camera whose lens leaves that range.
constexpr uint32_t kPassSpans = 2;  // spans of consecutive pixels (of one row) a pass hands out at most: one per half of the wave
constexpr uint32_t kPassRecBytes = 48;  // one waiting camera path: origin, direction, attenuation, RNG, entry | line | depth
#ifndef RTIOW_PASS_KEEP
#define RTIOW_PASS_KEEP 64  // records a wave keeps beyond its idle slots AT MOST (launch_path takes what the LDS has room for): a pass then
                           // runs on min(64, idle + records) lanes.  (Rounds 2-4: 32, which is what two 512-thread groups per CU, each
                           // with its own copy of the scene, left room for; the cover frame's passes then made 55 rays on average.
                           // Round 5: ONE group of 1024 threads around one copy -- 26 KB of LDS back -- and 64 records: every pass on
                           // all 64 lanes, 1.75 -> 1.5 million passes a frame; 32 / 36 / 40 records in two groups 5.71 / 5.69 / 5.63 ms,
                           // 48 / 64 in one group 5.56 / 5.52: profiles/r05_ab_log.txt.)
#endif
#ifndef RTIOW_PASS_MIN_LANES
#define RTIOW_PASS_MIN_LANES 32
#endif
constexpr uint32_t kPassKeep = RTIOW_PASS_KEEP;
#ifndef RTIOW_PASS_MIN_SPP
#define RTIOW_PASS_MIN_SPP 8
#endif
constexpr uint32_t kPassMinSpp = RTIOW_PASS_MIN_SPP;  // samples per pixel from which on camera rays take the pass (64 rays: <= 8 pixels;
                                                      // cover frame, pass off / on: 2 spp 0.84 / 1.20 ms, 4 spp 0.83 / 0.82, 8 spp 1.24 / 1.18)
constexpr uint32_t kPassMinLanes = RTIOW_PASS_MIN_LANES;  // camera rays a pass must be able to make (idle slots + records) to be run
static_assert(kPassKeep <= 64u && (64u - kPassKeep) * kPassRecBytes <= wave_item_bytes(false),
              "with pass_keep records of its own a wave's other records must fit the work-list area");

struct ConeAxis {  // per lane: the cone of the span this half of the wave looks at
    float ix, iy, iz, ax, ay, az;  // reciprocal axis direction, -O / direction
    float dnx, dny, dnz;           // unit axis
    float kappa, inv1mk;           // slope of the cone radius per unit of axis length; (1 + margin) / (1 - kappa)
    bool all;                      // degenerate cone: no cull
};

// (columns i_lo..i_hi of row j of the frame)
HDI ConeAxis cone_of_span(const PathArgs& a, const PersistArgs& g, uint32_t i_lo, uint32_t i_hi, uint32_t j) {
    const RtCamera& c = a.cam;
    const float uc = (0.5f * static_cast<float>(i_lo + i_hi) + 0.5f) * a.inv_wm1;
    const float vc = (static_cast<float>(j) + 0.5f) * a.inv_hm1;
    // half the span across, half a pixel up (g.h_len, g.v_len: |horizontal|, |vertical| with their margin)
    const float rho_t = fma_(0.5f * static_cast<float>(i_hi - i_lo + 1u) * a.inv_wm1, g.h_len, 0.5f * a.inv_hm1 * g.v_len);
    const float dx = fma_(vc, c.vertical[0], fma_(uc, c.horizontal[0], c.lower_left[0])) - c.origin[0];
    const float dy = fma_(vc, c.vertical[1], fma_(uc, c.horizontal[1], c.lower_left[1])) - c.origin[1];
    const float dz = fma_(vc, c.vertical[2], fma_(uc, c.horizontal[2], c.lower_left[2])) - c.origin[2];
    const float len2 = fma_(dz, dz, fma_(dy, dy, dx * dx));
    const float inv_len = cull_rsqrt(len2);  // (the axis need only be a unit vector to a few ulps: the cone's margins are 2^-8)
    ConeAxis o;
    o.dnx = dx * inv_len;
    o.dny = dy * inv_len;
    o.dnz = dz * inv_len;
    o.kappa = (g.lens_rho + rho_t) * inv_len;
    o.all = !(o.kappa < 0.125f);  // (also a NaN or an axis of length 0)
    o.inv1mk = 1.00390625f * cull_rcp(1.0f - o.kappa);
    o.ix = slab_rcp(o.dnx);
    o.iy = slab_rcp(o.dny);
    o.iz = slab_rcp(o.dnz);
    o.ax = -c.origin[0] * o.ix;
    o.ay = -c.origin[1] * o.iy;
    o.az = -c.origin[2] * o.iz;
    return o;
}

HDI bool cone_reaches(const PathArgs& a, const PersistArgs& g, const ConeAxis& c, const float4& mid, const float4& half) {
    const float ex = mid.x - a.cam.origin[0], ey = mid.y - a.cam.origin[1], ez = mid.z - a.cam.origin[2];
    const float proj = fma_(ez, c.dnz, fma_(ey, c.dny, ex * c.dnx));
    const float s_far = ((proj > 0.0f ? proj : 0.0f) + ((half.x + half.y) + half.z) + g.lens_rho) * c.inv1mk;
    const float rad = fma_(c.kappa, s_far, g.lens_rho) * 1.00390625f + g.abs_margin;
    const float4 fat = make_float4(half.x + rad, half.y + rad, half.z + rad, 0.0f);
    const float gap = slab_gap(mid, fat, c.ix, c.iy, c.iz, c.ax, c.ay, c.az);
    return c.all || !(gap < 0.0f);  // (a NaN reaches)
}

// One 32-bit mask, wave-uniform: bit k set = box first + k (of `boxes`, n of them) can be reached by a ray of one
// of the spans.  Lanes 0-31 look at one span and lanes 32-63 at the next: two spans per evaluation.
template <bool FLAT>
DI uint32_t cone_mask(const PathArgs& a, const PersistArgs& g, const float4* bounds, uint32_t base, uint32_t n, uint32_t first,
                      const ConeAxis& c0) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = first + (lane & 31u);
    const bool in = b < n;
    float4 mid, half;
    load_box<FLAT>(bounds, a, base + (in ? b : 0u), mid, half);  // (`base`: number of the level's first box in the list)
    bool reach = in && cone_reaches(a, g, c0, mid, half);
    const unsigned long long m = __ballot(reach);
    return static_cast<uint32_t>(m) | static_cast<uint32_t>(m >> 32);
}

// The same for one SPHERE {centre, r^2} (a slot of the LDS list): can a ray of the span be accepted by the exact test?
// binary32 evaluates the discriminant with an absolute error below ~22 eps |oc|^2, so an accepted ray passes
// within sqrt(r^2 + 64 eps |oc|^2) of the centre, and |oc| <= |e| + rho_L for a camera ray; the distance of the centre
// from the axis comes from the cross product (no cancellation: its error is a few eps |e|, against the absolute margin of
// 2^-16 of the coordinate range).  In front of the camera the axis counts as a line, behind it as the point O.
HDI bool cone_reaches_sphere(const PathArgs& a, const PersistArgs& g, const ConeAxis& c, const float4& s) {
    const float ex = s.x - a.cam.origin[0], ey = s.y - a.cam.origin[1], ez = s.z - a.cam.origin[2];
    const float proj = fma_(ez, c.dnz, fma_(ey, c.dny, ex * c.dnx));
    const float ee = fma_(ez, ez, fma_(ey, ey, ex * ex));
    const float cx = ey * c.dnz - ez * c.dny, cy = ez * c.dnx - ex * c.dnz, cz = ex * c.dny - ey * c.dnx;
    const float d2 = proj > 0.0f ? fma_(cz, cz, fma_(cy, cy, cx * cx)) : ee;
    const float rr = cull_sqrt(fma_(0x1p-17f, fma_(g.lens_rho, g.lens_rho, ee), s.w)) + g.abs_margin;
    const float s_far = ((proj > 0.0f ? proj : 0.0f) + rr + g.lens_rho) * c.inv1mk;
    const float lim = rr + (fma_(c.kappa, s_far, g.lens_rho) * 1.00390625f + g.abs_margin);
    return s.w >= 0.0f && (c.all || !(d2 > lim * lim));  // (padding slots have r^2 = -inf; a NaN reaches)
}

// exact test of the sphere in slot `slot` (wave-uniform: LDS broadcast) for the lane's own ray, branch-free
[[maybe_unused]] DI void exact_keyed_lockstep(const float4* slots, const SlotIndex* idx_map, uint32_t slot, const Path& p,
                             unsigned long long& key) {
    const float4 s = slots[slot];
    const uint32_t lo = (static_cast<uint32_t>(idx_map[slot]) << 16) | slot;
    const float ocx = p.o.x - s.x, ocy = p.o.y - s.y, ocz = p.o.z - s.z;
    const float hb = fma_(ocz, p.du.z, fma_(ocy, p.du.y, ocx * p.du.x));
    const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
    const float disc = fma_(hb, hb, -cc);
    const float sq = psqrt(disc);
    float root = -hb - sq;
    root = root > kTMin ? root : -hb + sq;
    // (a discriminant that is negative, -0 or NaN gives no hit, as in examine_keyed)
    const bool hit = static_cast<int32_t>(__float_as_uint(disc)) >= 0 && root > kTMin;
    const unsigned long long k2 = hit ? (static_cast<unsigned long long>(__float_as_uint(root)) << 32) | lo : ~0ull;
    key = k2 < key ? k2 : key;
}

// Closest hit of the camera rays of one pass (`active` lanes hold one each; the spans are wave-uniform).  Two culls,
// both wave-uniform: the cluster boxes the spans' cones reach, then -- lane by lane, one sphere each -- the members of
// those clusters (and the large spheres) the cones reach; what is left, typically the ground and a sphere or two, takes
// the exact test in lock-step.  Same test, same key, same minimum as trace_clustered for every sphere that can be hit.
template <bool SUPER, bool FLAT>
DI void primary_trace(const float4* slots, const SlotIndex* idx_map, const float4* bounds, const PathArgs& a,
                      const PersistArgs& g, const Path& p, bool active, const uint32_t (&span_col)[kPassSpans],
                      const uint32_t (&span_len)[kPassSpans], const uint32_t (&span_row)[kPassSpans], uint32_t n_spans, float& best, int& best_i,
                      uint32_t& best_o, uint32_t& n_tests, unsigned long long& dbg_trips) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t half = lane & 31u;
    unsigned long long key = ~0ull;
    const bool use_all = g.primary_all != 0u;
    const uint32_t q0 = lane < 32u || n_spans < 2u ? 0u : 1u;
    const ConeAxis c0 = cone_of_span(a, g, span_col[q0], span_col[q0] + span_len[q0] - 1u, span_row[q0]);
    // one sphere per lane of each half of the wave (slot_of(half): its slot, ~0u: none) against the cones; the spheres
    // that are reached, in lock-step against the rays
    auto spheres = [&](auto&& slot_of) {
        const uint32_t slot = slot_of(half);
        const bool in = slot != ~0u;
        const float4 s = slots[in ? slot : 0u];
        bool reach = in;
        if (!use_all) {
            reach = in && cone_reaches_sphere(a, g, c0, s);
            n_tests += in ? 1u : 0u;
        } else {
            reach = in && s.w >= 0.0f;
        }
        const unsigned long long m = __ballot(reach);
        uint32_t sm = static_cast<uint32_t>(m) | static_cast<uint32_t>(m >> 32);
        DBG_ADD(dbg_trips, lane == 0u ? __builtin_popcount(sm) : 0);
        while (sm) {
            const uint32_t bit = static_cast<uint32_t>(__builtin_ctz(sm));
            sm &= sm - 1u;
            exact_keyed_lockstep(slots, idx_map, __builtin_amdgcn_readfirstlane(slot_of(bit)), p, key);
            if (active) ++n_tests;
        }
    };
    // ---- the large spheres
    for (uint32_t j = 0; j < a.n_large; j += 32u) spheres([&](uint32_t h) { return j + h < a.n_large ? j + h : ~0u; });
    // ---- the clusters the cones reach, two at a time
    uint32_t held = ~0u;  // a reached cluster waiting for a partner
    auto cluster = [&](uint32_t c) {
        if (held == ~0u) {
            held = c;
            return;
        }
        const uint32_t ca = held;
        held = ~0u;
        spheres([&](uint32_t h) { return a.n_large_slots + (h < kClusterSize ? ca : c) * kClusterStride + (h & (kClusterSize - 1u)); });
    };
    static_assert(kClusterSize == 16u, "two clusters fill one half of the wave");
    if (!SUPER || a.n_super == 0u) {
        for (uint32_t g0 = 0; g0 < a.n_clusters; g0 += 32u) {
            uint32_t cm = 0xFFFFFFFFu;
            if (!use_all) {
                cm = cone_mask<FLAT>(a, g, bounds, 0u, a.n_clusters, g0, c0);
                n_tests += g0 + half < a.n_clusters ? 1u : 0u;
            }
            if (a.n_clusters - g0 < 32u) cm &= (1u << (a.n_clusters - g0)) - 1u;
            while (cm) {
                const uint32_t bit = static_cast<uint32_t>(__builtin_ctz(cm));
                cm &= cm - 1u;
                cluster(g0 + bit);
            }
        }
    } else {
        for (uint32_t s0 = 0; s0 < a.n_super; s0 += 32u) {
            uint32_t sm = 0xFFFFFFFFu;
            if (!use_all) {
                sm = cone_mask<FLAT>(a, g, bounds, a.n_clusters, a.n_super, s0, c0);
                n_tests += s0 + half < a.n_super ? 1u : 0u;
            }
            if (a.n_super - s0 < 32u) sm &= (1u << (a.n_super - s0)) - 1u;
            // the cluster boxes of four reached super-clusters at a time (4 x kSuperSize = 32 lanes per span)
            static_assert(kSuperSize == 8u, "four super-clusters fill one 32-bit mask");
            while (sm) {
                uint32_t sup[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    sup[u] = sm ? s0 + static_cast<uint32_t>(__builtin_ctz(sm)) : ~0u;
                    sm &= sm - 1u;  // (0 stays 0)
                }
                auto sup_of = [&](uint32_t part) { return part == 0u ? sup[0] : (part == 1u ? sup[1] : (part == 2u ? sup[2] : sup[3])); };
                const uint32_t my_sup = sup_of(half >> 3);
                const bool in = my_sup != ~0u;
                const uint32_t c = (in ? my_sup : 0u) * kSuperSize + (lane & 7u);
                bool reach = in;
                if (!use_all) {
                    float4 mid, hext;
                    load_box<FLAT>(bounds, a, c, mid, hext);
                    reach = in && cone_reaches(a, g, c0, mid, hext);
                    n_tests += in ? 1u : 0u;
                }
                const unsigned long long m = __ballot(reach);
                uint32_t cm = static_cast<uint32_t>(m) | static_cast<uint32_t>(m >> 32);
                while (cm) {
                    const uint32_t bit = static_cast<uint32_t>(__builtin_ctz(cm));
                    cm &= cm - 1u;
                    cluster(sup_of(bit >> 3) * kSuperSize + (bit & 7u));
                }
            }
        }
    }
    if (held != ~0u) {  // an odd one left
        const uint32_t ca = held;
        spheres([&](uint32_t h) { return h < kClusterSize ? a.n_large_slots + ca * kClusterStride + h : ~0u; });
    }
    const bool hit = active && key != ~0ull;
    best = hit ? __uint_as_float(static_cast<uint32_t>(key >> 32)) : __builtin_inff();
    best_i = hit ? static_cast<int>(key & 0xFFFFu) : -1;
    best_o = static_cast<uint32_t>(key >> 16) & 0xFFFFu;
}

#ifndef RTIOW_SMALL_MAX_THREADS
// largest group of the small-scene clustered variants: ONE group of 1024 threads per CU is their sixteen waves since round 5 (one copy of
// the scene in LDS instead of two: room for 64 pass records per wave, see RTIOW_PASS_KEEP; rounds 3-4: two groups of 512)
#define RTIOW_SMALL_MAX_THREADS 1024
#endif
#ifndef RTIOW_SMALL_WAVES_PER_EU
#define RTIOW_SMALL_WAVES_PER_EU 1  // (this compilation: no constraint.  The Makefile's second pass over this file, which makes the
                                    // small-scene clustered variants that are actually launched, sets 4: see the end of the file)
#endif
#ifndef RTIOW_LARGE_WAVES_PER_EU
#define RTIOW_LARGE_WAVES_PER_EU 3  // (4: A/B only -- what a fourth wave per SIMD would cost the large-scene variants in spilled registers)
#endif
#ifndef RTIOW_ACCEL_MAX_THREADS
#define RTIOW_ACCEL_MAX_THREADS 768
#endif
// The large-scene clustered variants want ~166 VGPRs: three waves per SIMD (<= 168), i.e. at most 768 threads per CU -- ONE
// group of 768 for a scene whose lists leave room for one copy only (C5's 4099 spheres: 87 KiB; round 1 stopped at 512: two
// waves per SIMD) -- held there by amdgpu_waves_per_eu.  The small-scene variants run FOUR waves per SIMD at 128 VGPRs, two
// groups of 512 per CU (rounds 1-2: three waves, three groups of 256 at ~160 registers): see the end of the file.
// tests/test_host_logic.py checks both budgets against the compiler's report.
constexpr int kAccelMaxThreads = RTIOW_ACCEL_MAX_THREADS;

// ---- The sparse loop at the end of a wave's frame (small-scene clustered kernel) -----------------------------------------------
// Once the queues are dry and a wave is down to kSparseParMax paths, nothing of the main loop's refill is of use to it
// any more -- no pool to fetch, no primary pass, no second slot: it gathers its paths in slot 0 once and ends its frame
// in a loop of its own, path-parallel sparse trace + one pass of the shade code.  What ends a small frame is the latency
// of ~50 such iterations (the fifty-bounce paths inside glass: 0.09 % of all paths, tools/path_lengths.py), so every
// instruction off them counts: 1/8 of the cover frame 1.45 -> 1.40 ms, a 1-spp frame 0.51 -> 0.48, the whole frame
// unchanged (8.21 / 8.27, within noise).
// (Round 3 also built the "express lane" VERDICT r2 proposed -- paths of 12+ segments handed, through a ring of LDS
// records, to a dedicated wave per workgroup that runs nothing but the sparse trace; frames identical, but the variant
// lost: profiles/r03_express_lane_ablation.txt; the code is in commit "Express lane v2".)
#ifndef RTIOW_TAIL_LOOP
#define RTIOW_TAIL_LOOP 1  // (-DRTIOW_TAIL_LOOP=0: A/B only)
#endif
#ifndef RTIOW_SPARSE_BATCHED
#define RTIOW_SPARSE_BATCHED 1  // (-DRTIOW_SPARSE_BATCHED=0: A/B only)
#endif
// FLAT (clustered kernels): the scene's cluster boxes share one interval along a.flat_axis, and the lock-step box tests
// leave that axis out (slab_gap_flat).
// COMPACT (round 5; large-scene clustered kernels only): FOUR waves per SIMD -- one group of 1024 threads at 128 registers -- around a scene
// whose lists leave 4.6 KB of LDS per wave instead of 5.9 (C5: 87 KB of lists): 32 accumulator entries and two line buffers per wave.
// Round 4 could not build it: at 128 registers the variant spilled 115 of them.  With the cold arguments out of the scalar registers
// (reload_path_args) it spills four under the default scheduler -- which is why these instantiations live in the whole-file compilation
// pass, not in the large-scene variants' own (iterative-ilp spills 44 at 128).
template <bool SHADE_LDS, bool ACCEL, bool FLAT = false, bool COMPACT = false>
__global__ __launch_bounds__(COMPACT ? 1024 : (ACCEL ? (SHADE_LDS ? RTIOW_SMALL_MAX_THREADS : kAccelMaxThreads) : 1024)) __attribute__((amdgpu_waves_per_eu(COMPACT ? 4 : (ACCEL && !SHADE_LDS ? RTIOW_LARGE_WAVES_PER_EU : RTIOW_SMALL_WAVES_PER_EU))))
void path_persistent_kernel(PathArgs a, PersistArgs g) {
#ifdef RTIOW_DEBUG_TIMELINE
    const unsigned long long tl_entry = wall_clock64();  // (the wave's first instruction: what the scene's staging costs -- VERDICT r4's start-of-tile question)
#endif
    static_assert(ACCEL || !FLAT, "only the clustered list has boxes");
    static_assert(!COMPACT || (ACCEL && !SHADE_LDS), "the compact per-wave area is for the large-scene clustered kernels");
    constexpr uint32_t kAccE = COMPACT ? kAccEntriesCompact : kAccEntries, kLineB = COMPACT ? kLineBufsCompact : kLineBufs;
    constexpr uint32_t kAccBytes = kAccE * kAccWords * 8u, kPixBytes = kAccE * 4u, kLineBytes = kLineB * (kChunkPix + kLineMetaWords) * 4u;
    // LDS: sphere list [g.n_pad float4] — the flat list, or (ACCEL) the clustered list's slots, then
    // (ACCEL) the slots' original indices [g.n_pad u32] and the boxes [2 (a.n_clusters + a.n_super) float4];
    // then (SHADE_LDS) two float4 of shading record per sphere; then the accumulator entries of every
    // wave of the workgroup (a wave allocates only from its own 64).
    // (small scenes only: the large-scene variant, at its 168 registers, spills two of them with the second copy of the
    // sparse trace, and the end of its frames -- seconds long -- does not matter)
    constexpr bool kSparseLoop = RTIOW_TAIL_LOOP != 0 && ACCEL && SHADE_LDS;
    // (No static __shared__ variable in this kernel: the dynamic area then starts at LDS address 0 and every address in it is a pure
    // offset -- the member reads of examine_cluster XOR a lane's byte offset with k * 16 and need no base added.  The workgroup's few
    // words of its own live in the dynamic area too, kGroupLdsBytes in front of the per-wave areas.)
    extern __shared__ float4 lds_spheres[];
    // The shader clock this frame ran at, from one wave's two clocks (shader cycles, and the constant 100 MHz counter): the stamps
    // are parked in LDS, not in registers.  (The fp32 peak a frame can be rated against is 157.3 TFLOP/s at 2.4 GHz;
    // under this kernel the part holds about 2.0: bench.py prints both.)
    SlotIndex* lds_cidx = reinterpret_cast<SlotIndex*>(lds_spheres + g.n_pad);  // (n_pad is a multiple of 16: the boxes behind stay 16-byte aligned)
    float4* lds_cbounds = reinterpret_cast<float4*>(lds_cidx + (ACCEL ? g.n_pad : 0u));
    // (boxes: centre + half extent, two float4 each; FLAT: without the flat axis, one float4 each)
    float4* lds_shade = lds_cbounds + (ACCEL ? (FLAT ? 1u : 2u) * (a.n_clusters + a.n_super) : 0u);
    // Behind the scene, one contiguous area per wave (g.wave_bytes of it): the accumulator entries of its pixels in flight {r, g, b,
    // samples done | cost}, the pixel of each entry, the line buffers of the chunks it is assembling, (ACCEL) the result keys and the
    // phase-2 work list(s) of trace_clustered, (ACCEL) pass_keep records of camera paths waiting for a slot -- all at fixed offsets
    // from ONE base.  (Rounds 1-3 laid every one of these out as an array over the workgroup's waves: seven base addresses to hold.)
    const uint32_t waves_in_group = blockDim.x / 64u;
    const uint32_t wave_in_group = threadIdx.x / 64u;
    unsigned char* lds_group = reinterpret_cast<unsigned char*>(lds_shade + (SHADE_LDS ? 2u * a.n : 0u));
    unsigned long long* wg_sums = reinterpret_cast<unsigned long long*>(lds_group);  // [3] paths, segments, tests of the waves that have left
    unsigned int* wg_left_p = reinterpret_cast<unsigned int*>(lds_group + 24);       // how many have
    unsigned long long* clk_start = reinterpret_cast<unsigned long long*>(lds_group + 32);  // [2] one wave's two clocks at its start
    if (threadIdx.x < 3u) wg_sums[threadIdx.x] = 0ull;
    if (threadIdx.x == 3u) *wg_left_p = 0u;
    if (threadIdx.x == 0u) {
        clk_start[0] = __builtin_readcyclecounter();
        clk_start[1] = wall_clock64();
    }
    unsigned char* lds_wave = lds_group + kGroupLdsBytes + wave_in_group * g.wave_bytes;
    unsigned long long* lds_acc = reinterpret_cast<unsigned long long*>(lds_wave);
    uint32_t* lds_pix = reinterpret_cast<uint32_t*>(lds_wave + kAccBytes);
    uint32_t* lds_line = reinterpret_cast<uint32_t*>(lds_wave + kAccBytes + kPixBytes);
    uint32_t* lds_line_meta = lds_line + kLineB * kChunkPix;  // per buffer {pixels done, pixels expected, cost (u64)}
    [[maybe_unused]] unsigned long long* lds_results = reinterpret_cast<unsigned long long*>(lds_wave + kAccBytes + kPixBytes + kLineBytes);
    [[maybe_unused]] uint16_t* lds_items = reinterpret_cast<uint16_t*>(lds_results + 128);
    // (the small-scene variants never have super-clusters: their records sit at a fixed offset too)
    [[maybe_unused]] float4* lds_pbuf = reinterpret_cast<float4*>(
        reinterpret_cast<unsigned char*>(lds_results) + wave_item_bytes(SHADE_LDS ? false : a.n_super != 0u));
    if (ACCEL) {
        for (uint32_t i = threadIdx.x; i < g.n_pad; i += blockDim.x) {
            lds_spheres[i] = a.cslots[i];
            lds_cidx[i] = static_cast<SlotIndex>(a.cidx[i]);  // (padding: 0xFFFF)
        }
        if (FLAT) {  // (a list of their own in HBM, clusters first: launch_path may have dropped the super level)
            for (uint32_t i = threadIdx.x; i < a.n_clusters + a.n_super; i += blockDim.x) lds_cbounds[i] = a.cbounds2[i];
        } else {
            for (uint32_t i = threadIdx.x; i < 2u * (a.n_clusters + a.n_super); i += blockDim.x) lds_cbounds[i] = a.cbounds[i];
        }
    } else {
        stage_spheres(a, lds_spheres, g.n_pad);
    }
    if (SHADE_LDS) {
        const float4* src = reinterpret_cast<const float4*>(a.shade);
        for (uint32_t i = threadIdx.x; i < 2u * a.n; i += blockDim.x) lds_shade[i] = src[i];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;

    Slot sl[kSlots];
#pragma unroll
    for (int r = 0; r < kSlots; ++r) {
        sl[r].active = false;
        sl[r].meta = 0u;
        sl[r].p.o = sl[r].p.du = sl[r].p.att = mk(0.0f, 0.0f, 0.0f);
    }
    // wave-uniform queue state (SGPRs)
    uint32_t pool_next = 0u, pool_end = 0u;  // the wave's pool: virtual pixel indices of queue pool_xcd not yet begun
    uint32_t pool_xcd = 0u, steal = 0u;      // the queue it came from; queues found dry so far
    // (the XCD this wave runs on and the tile's chunk count are worked out where a pool is fetched: hand_out)
    uint32_t cur_pix = 0xFFFFFFFEu, cur_entry = 0u;  // pixel being handed out (none yet) and its accumulator entry
    uint32_t cur_col = 0u, cur_row = 0u;     // ... its column and its row of the frame
    uint32_t cur_seq = ~0u, cur_chunk = 0u;  // position in the chunk sequence the wave is in, and the chunk there
    uint32_t cur_s = a.spp;                  // its next sample; == spp: no pixel open
    unsigned long long free_entries = kAccE == 64u ? ~0ull : (1ull << (kAccE & 63u)) - 1ull;  // accumulator entries not in use
    unsigned long long done_entries = 0ull;  // ... entries whose pixel has all its samples and waits to be resolved (resolve_done)
    uint32_t free_lines = (1u << kLineB) - 1u;  // line buffers not in use
    uint32_t whole_done = 0u;                // bit q: the whole-chunk part of queue q is known to be handed out
    uint32_t rest_done = 0u;                 // ... and the rest of it
    bool pool_owned = false;                 // the pool is whole chunks of the frame that only this wave renders
    bool pool_fine = g.fine_until != 0u;     // the wave's next small pool is a single pixel (the dear head of an ordered queue)
    uint32_t cur_line = 0u;                  // line buffer of the chunk being handed out, + 1 (0: its pixels go straight to the frame)
    [[maybe_unused]] uint32_t pass_n = 0u;   // (clustered) camera paths waiting in the wave's LDS records for an idle slot
    [[maybe_unused]] bool exhausted = false; // the global queue has been drained
#ifdef RTIOW_DEBUG_TIMELINE
    const unsigned long long tl_start = wall_clock64();
    // (one wave per workgroup: with one atomic per WAVE -- 4096 on one word, which the memory side performs one after the other -- every wave's
    // first load waited 30 us behind them, and the timeline showed a start-up cost that was its own)
    if (threadIdx.x == 0u) atomicMax(&a.counters->not_t0, ~tl_start);
    unsigned long long tl_dry = 0ull, tl_sparse = 0ull, tl_first_rays = 0ull, tl_first_pool = 0ull, tl_first_out = 0ull;  // (... first camera rays made, first pool fetched, first hand_out done)
    uint32_t tl_tail_iters = 0u, tl_sparse_iters = 0u, tl_sparse_paths = 0u, tl_starved = 0u, tl_live_at_dry = ~0u;
    uint32_t tl_deepest = 0u;            // (per lane) deepest path finished after dry
    unsigned long long tl_deep_end = 0ull;  // (per lane) when the last path of 40+ segments finished
#endif
    // Paths started, segments shaded, tests made: per lane -- or, paths and segments in the large-scene variants, per wave: scalars, where the
    // per-lane counts of rounds 1-4 were two registers all kernel long (the COMPACT variant spilled one).  The small-scene variants have the
    // registers and are short of scalars: the cover frame was 0.5 % slower with the scalar counts (profiles/r05_ab_log.txt).
    constexpr bool kWaveCounts = ACCEL && !SHADE_LDS;
    uint32_t n_paths = 0, n_segments = 0, n_tests = 0;
    uint32_t wave_paths = 0, wave_segments = 0;
#ifdef RTIOW_DEBUG_WAVE_POOLS
    uint32_t dbg_pools_drawn = 0u;
#endif
    [[maybe_unused]] uint32_t dbg_slow_trips = 0, dbg_cands = 0, dbg_iters = 0, dbg_sparse = 0;
    [[maybe_unused]] unsigned long long dbg_t_refill = 0, dbg_t_trace = 0, dbg_t_slow = 0, dbg_t_shade = 0;
    [[maybe_unused]] unsigned long long dbg_pass[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] const unsigned long long dbg_c0 = DBG_STAMP();
#ifdef RTIOW_DEBUG_COUNTERS
    const unsigned long long dbg_w0 = wall_clock64();
    if (lane == 0u) atomicMax(&a.counters->not_t0, ~dbg_w0);
    bool dbg_dry_seen = false;
    unsigned long long dbg_dry_at = 0ull, dbg_sp_ticks = 0ull;
    uint32_t dbg_tail_iters = 0u, dbg_sp_iters = 0u, dbg_sp_paths = 0u;
    unsigned long long dbg_tail_cyc[3] = {0ull, 0ull, 0ull};
#endif

    // ---- shade (used below, and by the primary pass inside the refill) ------
    // One segment of the path in q has been traced (best_i < 0: it left the scene).  Miss -> sky radiance into the
    // pixel's accumulator, hit -> scatter; a path that ends bumps its pixel's counter, the lane that completes a pixel
    // resolves and stores it.  r0, r1: the hit's shading record when the records are not in LDS.
    auto shade_one = [&](Slot& q, float hit_t, int hit_slot, uint32_t hit_orig, const float4& r0, const float4& r1) {
#ifndef RTIOW_SHADE_EARLY_OUT
#define RTIOW_SHADE_EARLY_OUT 1  // (-DRTIOW_SHADE_EARLY_OUT=0: A/B only)
#endif
        const unsigned long long shaded = __ballot(q.active);
        if (RTIOW_SHADE_EARLY_OUT && shaded == 0ull) return;  // (sparse iterations keep their paths in slot 0)
        if constexpr (kWaveCounts) wave_segments += static_cast<uint32_t>(__popcll(shaded));
        bool finished = false;
        if (q.active) {
            if constexpr (!kWaveCounts) ++n_segments;
            unsigned long long* acc = lds_acc + meta_entry(q.meta) * kAccWords;
            if (hit_slot < 0) {
                const f3 rad = sky_radiance(q.p);
                atomicAdd(acc + 0, to_fixed(rad.x));  // ds_add_u64: order-independent integer sum
                atomicAdd(acc + 1, to_fixed(rad.y));
                atomicAdd(acc + 2, to_fixed(rad.z));
                finished = true;
            } else {
                const float4 geo = lds_spheres[hit_slot];
                ShadeRec m;
                if (SHADE_LDS) {
                    const float4 m0 = lds_shade[2u * hit_orig], m1 = lds_shade[2u * hit_orig + 1u];
                    m.albedo[0] = m0.x; m.albedo[1] = m0.y; m.albedo[2] = m0.z; m.param = m0.w;
                    m.inv_r = m1.x; m.kind = __float_as_uint(m1.y);
                } else {
                    m.albedo[0] = r0.x; m.albedo[1] = r0.y; m.albedo[2] = r0.z; m.param = r0.w;
                    m.inv_r = r1.x; m.kind = __float_as_uint(r1.y);
                }
                if (!scatter(mk(geo.x, geo.y, geo.z), m, hit_t, q.p)) {
                    finished = true;  // absorbed: radiance 0
                } else {
                    q.meta += 1u << kMetaDepthShift;
                    if (meta_depth(q.meta) >= a.max_depth) finished = true;  // depth exhausted: radiance 0
                }
            }
        }
        // A finished sample bumps its pixel's counter; the sample that completes the pixel marks its accumulator entry as done
        // (resolve_done below turns done entries into pixels of the frame, several at a time).
        bool completed = false;
        if (finished) {
#ifdef RTIOW_DEBUG_TIMELINE
            if (tl_dry != 0ull && meta_depth(q.meta) + 1u > tl_deepest) tl_deepest = meta_depth(q.meta) + 1u;
            if (meta_depth(q.meta) + 1u >= 40u) tl_deep_end = wall_clock64();
#ifdef RTIOW_DEBUG_DEPTH_HIST
            // (a build of its own, EXTRA=-DRTIOW_DEBUG_DEPTH_HIST on top of `make tl`: atomics on a handful of words -- a frame takes three times
            // as long with them, from 8 segments on; with one per path, 0.9 s -- so the timeline build proper does without)
            if (meta_depth(q.meta) + 1u >= 8u) atomicAdd(&a.counters->tl_depth_hist[meta_depth(q.meta) + 1u < 63u ? meta_depth(q.meta) + 1u : 63u], 1ull);
#endif
#endif
            q.active = false;
            unsigned long long* acc = lds_acc + meta_entry(q.meta) * kAccWords;
            // low half: samples done; high half: the segments they took (<= 65535 each, <= 65536 samples)
            // (A long path counts kLongWeight-fold: what ends a frame is not the work of its last chunks but the LENGTH of
            // the paths born in them -- fifty bounces at one iteration each -- so chunks in which long paths occur are to go
            // out first whatever their average; the sum only orders the chunks of the next frame.)
            const uint32_t segs1 = meta_depth(q.meta) < 0xFFFEu ? meta_depth(q.meta) + 1u : 0xFFFFu;
            const uint32_t segs = segs1 > kLongFrom ? (segs1 * kLongWeight < 0xFFFFu ? segs1 * kLongWeight : 0xFFFFu) : segs1;
            uint32_t one = 1u;  // (made here, as fresh64's words are: the pair's constant half otherwise lives -- in the COMPACT variant: in scratch -- all kernel long)
            asm volatile("" : "+v"(one));
            const unsigned long long before = atomicAdd(acc + 3, one | (static_cast<unsigned long long>(segs) << 32));
            completed = static_cast<uint32_t>(before) + 1u == a.spp;
        }
        unsigned long long done_mask = __ballot(completed);  // (0-2 lanes per call at a hundred samples per pixel)
        while (done_mask != 0ull) {
            const int l = __builtin_ctzll(done_mask);
            done_mask &= done_mask - 1ull;
            done_entries |= 1ull << meta_entry(__builtin_amdgcn_readlane(q.meta, l));
        }
    };
    // Done entries -> pixels of the frame: lane e takes entry e.  All adds to an entry were issued by earlier LDS instructions of this
    // wave, so the sums are final; resolve, quantise and store are ~125 vector instructions, and rounds 1-3 ran them in the lane whose
    // sample completed the pixel, at once -- one or two lanes of 64, in every fourth pass of the shade code: 3 % of the kernel's
    // instructions for 1 / 64 of their worth.  Now the entries wait (the main loop calls this when kResolveBatch of them have
    // gathered, hand_out when it runs out of entries, the wave when it is through) and go eight and more at a time.
    auto resolve_done = [&]() {
        if (done_entries == 0ull) return;
        const PathArgs ca = reload_path_args();  // (cold: the frame's pointers, pitch and width, the accumulators, the quantiser)
        const uint32_t total_pix = reload_persist_args().total_pix;
        const bool mine = ((done_entries >> lane) & 1ull) != 0ull;
        bool line_full = false;  // this lane's pixel was the last of a line buffer
        uint32_t done_pix = 0u, my_line = 0u;
        if (mine) {
            unsigned long long* acc = lds_acc + lane * kAccWords;
            const uint32_t where = lds_pix[lane];  // (written when the pixel was opened: hand_out)
            done_pix = where & ((1u << kPixLineShift) - 1u);
            my_line = where >> kPixLineShift;
            const unsigned long long cost = acc[3] >> 32;  // what the pixel's samples cost (weighted segments: see shade_one)
            const uint32_t colour = close_pixel(ca, done_pix, acc[0], acc[1], acc[2]);
            if (my_line == 0u) {
                const uint32_t lr = pixel_row(ca, done_pix), i = done_pix - lr * ca.width;
                ca.dst[static_cast<size_t>(lr) * ca.dst_stride + i] = colour;
            } else {  // a pixel of a chunk this wave renders alone: into the line buffer
                lds_line[(my_line - 1u) * kChunkPix + done_pix % kChunkPix] = colour;
                uint32_t* meta = lds_line_meta + kLineMetaWords * (my_line - 1u);
                atomicAdd(reinterpret_cast<unsigned long long*>(meta + 2), cost);  // the chunk's cost, in LDS
                const uint32_t done = atomicAdd(meta, 1u);
                line_full = done + 1u == meta[1];
            }
            // what this pixel cost, for the next frame's chunk order (a global atomic leaves the L2 as a 64-byte
            // memory-side request: whole chunks sum theirs in LDS and report once, with the line)
            // ... and of the pixels handed out one by one, every kCostSample-th speaks for its neighbours -- unless a long
            // path ended in this one: those are what the order is for, and too rare to be sampled
            if (ca.chunk_cost != nullptr && my_line == 0u) {
                if (cost >= static_cast<unsigned long long>(ca.spp) * kLongFrom + kLongFrom * kLongWeight)  // (one long path at least)
                    atomicAdd(ca.chunk_cost + done_pix / kChunkPix, cost);
                else if ((done_pix & (kCostSample - 1u)) == 0u)
                    atomicAdd(ca.chunk_cost + done_pix / kChunkPix, static_cast<unsigned long long>(kCostSample) * cost);
            }
        }
        // a pixel that filled its line buffer has the line stored -- up to 32 consecutive pixels, 128 bytes, the whole line of the
        // frame in one store.  (A wave's LDS operations are performed in order: the colours written above are there.)
        unsigned long long full = __ballot(line_full);
        while (full != 0ull) {
            const int l = __builtin_ctzll(full);
            full &= full - 1ull;
            const uint32_t line = __builtin_amdgcn_readlane(my_line, l) - 1u;
            const uint32_t first = __builtin_amdgcn_readlane(done_pix, l) / kChunkPix * kChunkPix;
            const uint32_t count = total_pix - first < kChunkPix ? total_pix - first : kChunkPix;
            if (lane < count) {
                const uint32_t pix = first + lane;
                const uint32_t lr = pixel_row(ca, pix), i = pix - lr * ca.width;  // (a chunk may run over the end of a row)
                ca.dst[static_cast<size_t>(lr) * ca.dst_stride + i] = lds_line[line * kChunkPix + lane];
            }
            if (ca.chunk_cost != nullptr && lane == 0u)  // this wave rendered the whole chunk: a plain store
                ca.chunk_cost[first / kChunkPix] = *reinterpret_cast<const unsigned long long*>(lds_line_meta + kLineMetaWords * line + 2u);
            free_lines |= 1u << line;
        }
        free_entries |= done_entries;  // the entries return to the wave
        done_entries = 0ull;
    };
    [[maybe_unused]] bool to_sparse_loop = false;  // the wave leaves the main loop for the sparse loop below
    for (;;) {
        [[maybe_unused]] const unsigned long long t0 = DBG_STAMP();
        // ---- refill ---------------------------------------------------------
        // Hands out the next `want` samples of the wave's pool, pixel by pixel: on_range(first, n, pixel, column, row, entry, sample)
        // is told that the idle slots numbered first .. first+n-1 get samples sample .. sample+n-1 of that pixel (whose
        // line buffer is cur_line); it may refuse them (false), which ends the call.  Returns how many were handed out
        // (fewer than `want` once the queues are dry or the accumulator entries are all in use).
        auto hand_out = [&](uint32_t want, auto&& on_range) -> uint32_t {
            uint32_t served = 0u;  // wave-uniform
            while (served < want) {  // one trip per pixel touched (1-2 unless spp is tiny)
                if (cur_s == a.spp) {  // open the next pixel of the pool
                    // (cold: the queue's constants, the tile's shape, the chunk order -- read once per pixel opened)
                    const PathArgs ca = reload_path_args();
                    const PersistArgs cg = reload_persist_args();
                    const uint32_t n_chunks = (cg.total_pix + kChunkPix - 1u) / kChunkPix;
                    if (pool_next == pool_end) {
                        const uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;  // HW_REG_XCC_ID[3:0]
                        // Pool fetch.  The tile's pixels are cut into chunks of kChunkPix consecutive pixels dealt
                        // round-robin to eight queues, one per XCD: a wave draws from the queue of the XCD it runs
                        // on, so the 4-byte stores that complete a 128-byte line of the frame all come from one L2
                        // and merge there.  An XCD whose queue is dry steals from the next one.
                        bool fetched = false;
#ifdef RTIOW_DEBUG_WAVE_POOLS  // (tools/sparse_latency.py: a wave draws this many pools and no more)
                        if (dbg_pools_drawn >= RTIOW_DEBUG_WAVE_POOLS) steal = 8u;
                        ++dbg_pools_drawn;
#endif
                        while (steal < 8u && !fetched) {
                            const uint32_t xq = (xcc + steal) & 7u;
                            const uint32_t vsize = ((n_chunks + 7u - xq) / 8u) * kChunkPix;  // virtual pixels of queue xq
                            // Its first part -- all but the last g.chunk_until pixels -- is handed out in WHOLE chunks (32 pixels
                            // = one 128-byte line of the frame): the wave renders every pixel of the line itself, assembles it in
                            // LDS and writes it with one store.  That part has a head of its own and a fetch is one atomic add, no
                            // look at the head first (a look and an add are 64 bytes of fabric traffic each: with one queue for
                            // everything the looks alone were 1.9 MB per cover frame, half of what the frame itself weighs).
                            const uint32_t wsize = vsize > cg.chunk_until ? (vsize - cg.chunk_until) / cg.chunk_pool * cg.chunk_pool : 0u;
                            if (wsize != 0u && ((whole_done >> xq) & 1u) == 0u) {
                                uint32_t got = 0u;
                                if (lane == 0u) got = atomicAdd(&ca.counters->xcd_chunk[xq].next, cg.chunk_pool);
                                got = __builtin_amdgcn_readfirstlane(got);
                                if (got < wsize) {
                                    pool_next = got;
                                    pool_end = got + cg.chunk_pool;  // (<= wsize: both are multiples of the pool)
                                    pool_xcd = xq;
                                    pool_owned = true;
                                    fetched = true;
#ifdef RTIOW_DEBUG_TIMELINE
                                    if (xq == 0u && lane == 0u && got * 8u / vsize != (got + cg.chunk_pool) * 8u / vsize) {
                                        const unsigned long long t0w = ~__hip_atomic_load(&a.counters->not_t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        a.counters->tl_progress[(got + cg.chunk_pool) * 8u / vsize] = static_cast<unsigned int>(wall_clock64() - t0w);
                                    }
#endif
                                    break;
                                }
                                whole_done |= 1u << xq;  // (heads only grow)
                            }
                            // The rest of the queue -- by then every wave still holds half a chunk on average, which this part
                            // has to balance -- goes in pools of a few pixels, each pixel stored when it completes.  One atomic
                            // add per pool here too.  (Round 1 looked at the head first and shrank the pools towards the end of
                            // the queue -- "guided" sizes.  Pools are at most a handful of pixels anyway: with the shrinking
                            // switched off the cover frame and a 1/8 tile took the same time, and every look is 64 bytes of fabric
                            // traffic, 1.3 MB per frame.)
                            const uint32_t rest = vsize - wsize;
                            if (((rest_done >> xq) & 1u) == 0u) {
                                // The head of an ORDERED queue holds the dearest pixels of the frame -- the rim of the glass ball: a
                                // hundred samples that nearly all bounce fifty times -- and a pool is one wave's work whatever it holds:
                                // on a small frame four such pixels in one wave's hands are the whole frame's time (one eighth of the cover
                                // frame, tile 5: 1.60 ms with pools of 4, 2.0 with 8, 1.36 with 1; tools/tile_ranks.py).  So the first
                                // g.fine_until pixels of a queue go out an iteration's worth of samples at a time -- one pixel at 100 spp
                                // (judged by the wave's last fetch: no look at the head).
                                const uint32_t k = pool_fine ? cg.fine_pix : cg.pool_pix;
                                uint32_t got = 0u;
                                if (lane == 0u) got = atomicAdd(&ca.counters->xcd_head[xq].next, k);
                                got = __builtin_amdgcn_readfirstlane(got);
                                pool_fine = wsize + got + k < cg.fine_until;
#ifdef RTIOW_DEBUG_TIMELINE
                                if (xq == 0u && lane == 0u && got < rest && (wsize + got) * 8u / vsize != (wsize + got + k) * 8u / vsize) {
                                    const unsigned long long t0w = ~__hip_atomic_load(&a.counters->not_t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    a.counters->tl_progress[(wsize + got + k) * 8u / vsize] = static_cast<unsigned int>(wall_clock64() - t0w);
                                }
#endif
                                if (got < rest) {
                                    pool_next = wsize + got;
                                    pool_end = rest - got < k ? vsize : wsize + got + k;
                                    pool_xcd = xq;
                                    pool_owned = false;
                                    fetched = true;
                                    break;
                                }
                                rest_done |= 1u << xq;
                            }
                            ++steal;  // this queue is dry for good: heads only grow
                        }
                        if (!fetched) {
                            exhausted = true;
                            TL_MARK(tl_dry);
#ifdef RTIOW_DEBUG_COUNTERS
                            if (!dbg_dry_seen && lane == 0u) {
                                const unsigned long long t0w = ~__hip_atomic_load(&a.counters->not_t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const unsigned long long b = (wall_clock64() - t0w) / 12500ull;
                                atomicAdd(&a.counters->hist_dry[b > 31ull ? 31ull : b], 1u);
                            }
                            if (!dbg_dry_seen) dbg_dry_at = wall_clock64();
                            dbg_dry_seen = true;
#endif
                            break;
                        }
                    }
                    TL_MARK(tl_first_pool);
                    // virtual index of queue pool_xcd -> position in the chunk sequence -> chunk -> pixel of the tile
                    const uint32_t seq = (pool_next / kChunkPix) * 8u + pool_xcd;
                    if (seq != cur_seq) {  // (wave-uniform: one scalar load per chunk entered)
                        cur_seq = seq;
                        // (the order is stored queue by queue, so that an XCD reads its own eighth of it and no more)
                        cur_chunk = ca.chunk_order != nullptr
                                        ? __builtin_amdgcn_readfirstlane(ca.chunk_order[chunk_order_slot(seq, n_chunks)])
                                        : seq;
                        cur_line = 0u;
                        if (pool_owned && free_lines != 0u) {  // (no buffer free: this chunk's pixels go straight to the frame)
                            const uint32_t line = static_cast<uint32_t>(__builtin_ctz(free_lines));
                            free_lines &= free_lines - 1u;
                            cur_line = line + 1u;
                            const uint32_t first = cur_chunk * kChunkPix;
                            const uint32_t expect = cg.total_pix - first < kChunkPix ? cg.total_pix - first : kChunkPix;  // ragged last chunk
                            if (lane < kLineMetaWords) lds_line_meta[kLineMetaWords * line + lane] = lane == 1u ? expect : 0u;
                        }
                    }
                    const uint32_t pix = cur_chunk * kChunkPix + pool_next % kChunkPix;
                    if (pix >= cg.total_pix) {  // the ragged end of the last chunk
                        ++pool_next;
                        continue;
                    }
                    if (free_entries == 0ull) resolve_done();  // (entries that only wait to be stored)
                    if (free_entries == 0ull) {  // 64 pixels in flight: wait for one to finish
#ifdef RTIOW_DEBUG_TIMELINE
                        ++tl_starved;
#endif
                        break;
                    }
                    cur_entry = static_cast<uint32_t>(__builtin_ctzll(free_entries));
                    free_entries &= free_entries - 1ull;
                    // column and row (of the frame) of the pixel: by one division for the first pixel of a pool or a row, by counting
                    // for those that follow it (rounds 1-3: two or three divisions per lane and pass, a dozen instructions each)
                    if (pix == cur_pix + 1u && cur_col + 1u < ca.width) {
                        ++cur_col;
                    } else {
                        const uint32_t lr = pixel_row(ca, pix);
                        cur_col = pix - lr * ca.width;
                        cur_row = tile_global_row(ca, lr);
                    }
                    cur_pix = pix;
                    ++pool_next;
                    cur_s = 0u;
                    if (lane < kAccWords) lds_acc[cur_entry * kAccWords + lane] = fresh64(0u);
                    if (lane == kAccWords) lds_pix[cur_entry] = pix | (cur_line << kPixLineShift);
                }
                const uint32_t n = want - served < a.spp - cur_s ? want - served : a.spp - cur_s;
                if (!on_range(served, n, cur_pix, cur_col, cur_row, cur_entry, cur_s)) break;  // (the pixel stays open)
                cur_s += n;
                served += n;
            }
            return served;
        };
        bool any_active = false;
        if constexpr (ACCEL) {
            // Idle slots are filled with camera paths that have already taken their first segment: the primary pass
            // makes up to 64 camera rays at a time (one per lane, consecutive samples of the pool's pixels), traces them
            // against the clusters their pixels' cones reach, shades them, and leaves the paths that go on in LDS
            // records, where idle slots pick them up.  A pass makes up to pass_keep paths more than there are idle
            // slots, so that it runs on (nearly) all 64 lanes; the surplus waits in the wave's records for the next
            // iteration.  Records pass_keep.. live in the trace's work-list area (idle now) and are always taken first.
            float4* scratch4 = reinterpret_cast<float4*>(lds_results);
            auto record = [&](uint32_t k) { return k < g.pass_keep ? lds_pbuf + 3u * k : scratch4 + 3u * (k - g.pass_keep); };
            for (; !SHADE_LDS || g.use_pass != 0u;) {  // (large scenes: always, see below)
                const unsigned long long idle0 = __ballot(!sl[0].active), idle1 = kSlots > 1 ? __ballot(!sl[kSlots - 1].active) : 0ull;
                const uint32_t n_idle0 = static_cast<uint32_t>(__popcll(idle0));
                const uint32_t n_idle = n_idle0 + static_cast<uint32_t>(__popcll(idle1));
                if (n_idle == 0u) break;
                if (pass_n == 0u) {
                    // (a pass for a handful of camera rays costs as much as one for 64: with fewer records than that the
                    // idle slots wait until there are enough of them -- unless the wave has nothing else to do)
                    // (Only in the large-scene variant: small scenes always have room for their 32 records, and the mere
                    // presence of this test cost the cover frame 3 % -- 8.63 -> 8.88 ms -- through the code around it.)
                    if (!SHADE_LDS && n_idle < g.pass_min_idle && __ballot(sl[0].active || sl[kSlots - 1].active) != 0ull) break;
                    Slot ps;
                    ps.active = false;
                    ps.meta = 0u;
                    uint32_t gen_col = 0u, gen_row = 0u;
                    ps.p.o = ps.p.du = ps.p.att = mk(0.0f, 0.0f, 0.0f);
                    uint32_t gen_s = 0u;
                    // the spans of consecutive pixels (of one row) the pass hands out: wave-uniform
                    uint32_t span_col[kPassSpans], span_len[kPassSpans], span_row[kPassSpans];  // first column, pixels, row of the frame
#pragma unroll
                    for (uint32_t k = 0; k < kPassSpans; ++k) span_col[k] = span_len[k] = span_row[k] = 0u;
                    uint32_t n_spans = 0u, last_pix = 0u, last_col = 0u;
                    const uint32_t want = n_idle + g.pass_keep < g.pass_cap ? n_idle + g.pass_keep : g.pass_cap;
                    [[maybe_unused]] const unsigned long long th0 = DBG_STAMP();
                    const uint32_t granted = hand_out(want, [&](uint32_t first, uint32_t n, uint32_t pix, uint32_t col, uint32_t row, uint32_t entry, uint32_t s0) {
                        // (wave-uniform) the pixel continues the open span (the next pixel of the same row), or opens the next one -- a
                        // pass stops at the third: lanes 0-31 and 32-63 look at one span each when the cones are tested
                        const bool extends = n_spans != 0u && pix == last_pix + 1u && col == last_col + 1u;
                        if (!extends && n_spans == kPassSpans) return false;
                        if (extends) {
                            if (n_spans == 1u) ++span_len[0]; else ++span_len[1];
                        } else {
                            if (n_spans == 0u) { span_col[0] = col; span_len[0] = 1u; span_row[0] = row; }
                            else { span_col[1] = col; span_len[1] = 1u; span_row[1] = row; }
                            ++n_spans;
                        }
                        last_pix = pix;
                        last_col = col;
                        if (lane - first < n) {  // (unsigned: first <= lane < first + n)
                            gen_col = col;
                            gen_row = row;
                            ps.meta = meta_of(entry, cur_line);
                            gen_s = s0 + (lane - first);
                        }
                        return true;
                    });
                    TL_MARK(tl_first_out);
                    if (granted == 0u) break;  // queues dry, or all accumulator entries in use (then paths are in flight)
                    // (cold: the camera, the frame's size and seed, the cone test's margins -- read once per pass; what the per-segment code
                    // keeps in registers anyway comes from there)
                    PathArgs pa = reload_path_args();
                    pa.n_large = a.n_large; pa.n_large_slots = a.n_large_slots; pa.n_clusters = a.n_clusters; pa.n_super = a.n_super;
                    pa.flat_axis = a.flat_axis; pa.flat_mid = a.flat_mid; pa.flat_half = a.flat_half;
                    const PersistArgs pg = reload_persist_args();
                    [[maybe_unused]] const unsigned long long tp0 = DBG_STAMP();
                    DBG_ADD(dbg_pass[0], lane == 0u ? 1u : 0u);
                    DBG_ADD(dbg_pass[6], lane == 0u ? tp0 - th0 : 0ull);  // hand_out
                    DBG_ADD(dbg_pass[1], lane == 0u ? granted : 0u);
                    ps.active = lane < granted;
                    if (ps.active) {
                        camera_path(pa, gen_col, gen_row, pa.sample_offset + gen_s, ps.p);
                        if constexpr (!kWaveCounts) ++n_paths;
                    }
                    if constexpr (kWaveCounts) wave_paths += granted;
                    DBG_ADD(dbg_pass[7], lane == 0u ? DBG_STAMP() - tp0 : 0ull);  // camera_path
                    TL_MARK(tl_first_rays);
                    float pb;
                    int pb_i;
                    uint32_t pb_o;
                    primary_trace<!SHADE_LDS, FLAT>(lds_spheres, lds_cidx, lds_cbounds, pa, pg, ps.p, ps.active, span_col, span_len, span_row, n_spans,
                                              pb, pb_i, pb_o, n_tests, dbg_pass[2]);
                    float4 pr0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pr1 = pr0;
                    if (!SHADE_LDS && ps.active && pb_i >= 0) {
                        const float4* src = reinterpret_cast<const float4*>(a.shade + pb_o);
                        pr0 = src[0];
                        pr1 = src[1];
                    }
                    [[maybe_unused]] const unsigned long long tp1 = DBG_STAMP();
                    shade_one(ps, pb, pb_i, pb_o, pr0, pr1);
                    const unsigned long long going = __ballot(ps.active);
                    pass_n = static_cast<uint32_t>(__popcll(going));
                    DBG_ADD(dbg_pass[3], lane == 0u ? pass_n : 0u);
                    DBG_ADD(dbg_pass[4], lane == 0u ? DBG_STAMP() - tp0 : 0ull);
                    DBG_ADD(dbg_pass[5], lane == 0u ? DBG_STAMP() - tp1 : 0ull);
                    if (ps.active) {
                        float4* rec = record(lane_rank(going));
                        rec[0] = make_float4(ps.p.o.x, ps.p.o.y, ps.p.o.z, ps.p.du.x);
                        rec[1] = make_float4(ps.p.du.y, ps.p.du.z, ps.p.att.x, ps.p.att.y);
                        rec[2] = make_float4(ps.p.att.z, __uint_as_float(ps.p.rng.state), __uint_as_float(ps.meta), 0.0f);  // (depth 1)
                    }
                    if (pass_n == 0u) continue;  // every ray left the scene (sky): the slots are as idle as before
                }
                // the last `give` records go to the idle slots, numbered across both slots
                // (a wave's LDS operations are performed in order: the records are there)
                [[maybe_unused]] const unsigned long long td0 = DBG_STAMP();
                const uint32_t give = pass_n < n_idle ? pass_n : n_idle;
                uint32_t my_idx[kSlots];
                my_idx[0] = lane_rank(idle0);
                if (kSlots > 1) my_idx[kSlots - 1] = n_idle0 + lane_rank(idle1);
#pragma unroll
                for (int r = 0; r < kSlots; ++r) {
                    Slot& q = sl[r];
                    if (!q.active && my_idx[r] < give) {
                        const float4* rec = record(pass_n - give + my_idx[r]);
                        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                        q.p.o = mk(r0.x, r0.y, r0.z);
                        q.p.du = mk(r0.w, r1.x, r1.y);
                        q.p.att = mk(r1.z, r1.w, r2.x);
                        q.p.rng = Pcg(__float_as_uint(r2.y));
                        q.meta = __float_as_uint(r2.z);
                        q.active = true;
                    }
                }
                pass_n -= give;
                DBG_ADD(dbg_pass[8], lane == 0u ? DBG_STAMP() - td0 : 0ull);  // records -> slots
                DBG_ADD(dbg_pass[9], lane == 0u ? 1u : 0u);
                if (pass_n != 0u) break;  // (more records than idle slots: the slots are full)
            }
            if (SHADE_LDS && g.use_pass == 0u) {
                // Few samples per pixel (launch_path: fewer than kPassMinSpp): the camera rays a wave starts together
                // belong to dozens of pixels, their common cone is wide and culls little -- they go into the slots
                // untraced and take the general trace like any other ray.  (Small scenes only: the large-scene variant
                // has no registers to spare for a second way in, and its general trace is dear enough -- three list
                // stages -- for the pass to pay even with a wide cone.)  The idle slots are numbered across both slots,
                // lane k generates the camera path of number k -- ONE pass of the camera code instead of one per slot
                // -- and leaves it in the wave's LDS scratch, where the lane that owns slot number k picks it up.
                const unsigned long long idle0 = __ballot(!sl[0].active), idle1 = kSlots > 1 ? __ballot(!sl[kSlots - 1].active) : 0ull;
                const uint32_t n_idle0 = static_cast<uint32_t>(__popcll(idle0));
                const uint32_t n_idle = n_idle0 + static_cast<uint32_t>(__popcll(idle1));
                uint32_t left = n_idle;
                while (left != 0u) {  // (at most two trips: 64 lanes per trip)
                    const uint32_t done = n_idle - left;
                    uint32_t my_idx[kSlots];  // (unsigned: < 64 = this trip)
                    my_idx[0] = lane_rank(idle0) - done;
                    if (kSlots > 1) my_idx[kSlots - 1] = n_idle0 + lane_rank(idle1) - done;
                    bool got[kSlots];
#pragma unroll
                    for (int r = 0; r < kSlots; ++r) got[r] = false;
                    uint32_t gen_col = 0u, gen_row = 0u, gen_s = 0u;
                    const uint32_t want = left < 64u ? left : 64u;
                    const uint32_t granted = hand_out(want, [&](uint32_t first, uint32_t n, uint32_t, uint32_t col, uint32_t row, uint32_t entry, uint32_t s0) {
                        if (lane - first < n) {  // (unsigned: first <= lane < first + n)
                            gen_col = col;
                            gen_row = row;
                            gen_s = s0 + (lane - first);
                        }
#pragma unroll
                        for (int r = 0; r < kSlots; ++r)
                            if (!sl[r].active && my_idx[r] - first < n) {
                                sl[r].meta = meta_of(entry, cur_line);  // (depth 0)
                                got[r] = true;
                            }
                        return true;
                    });
                    float4* rec = scratch4;  // [64] {o, d.x} then [64] {d.y, d.z, rng, -}
                    if (lane < granted) {
                        Path np;
                        const PathArgs pa = reload_path_args();  // (cold: the camera)
                        camera_path(pa, gen_col, gen_row, pa.sample_offset + gen_s, np);
                        rec[lane] = make_float4(np.o.x, np.o.y, np.o.z, np.du.x);
                        rec[64u + lane] = make_float4(np.du.y, np.du.z, __uint_as_float(np.rng.state), 0.0f);
                    }
#pragma unroll
                    for (int r = 0; r < kSlots; ++r) {
                        Slot& q = sl[r];
                        if (got[r]) {  // (a wave's LDS operations are performed in order: the record is there)
                            const float4 r0 = rec[my_idx[r]], r1 = rec[64u + my_idx[r]];
                            q.p.o = mk(r0.x, r0.y, r0.z);
                            q.p.du = mk(r0.w, r1.x, r1.y);
                            q.p.rng = Pcg(__float_as_uint(r1.z));
                            q.p.att = mk(1.0f, 1.0f, 1.0f);
                            q.active = true;
                            ++n_paths;
                        }
                    }
                    if (granted < want) break;
                    left -= granted;
                }
            }
            any_active = sl[0].active || sl[kSlots - 1].active;
        } else {
            const unsigned long long idle0 = __ballot(!sl[0].active), idle1 = kSlots > 1 ? __ballot(!sl[kSlots - 1].active) : 0ull;
#pragma unroll
            for (int r = 0; r < kSlots; ++r) {  // slot by slot
                Slot& q = sl[r];
                const unsigned long long mask = r == 0 ? idle0 : idle1;
                const uint32_t rank = lane_rank(mask);
                uint32_t my_s = 0u, my_col = 0u, my_row = 0u;
                bool got_sample = false;
                hand_out(static_cast<uint32_t>(__popcll(mask)), [&](uint32_t first, uint32_t n, uint32_t, uint32_t col, uint32_t row, uint32_t entry, uint32_t s0) {
                    if (!q.active && rank - first < n) {
                        my_col = col;
                        my_row = row;
                        q.meta = meta_of(entry, cur_line);  // (depth 0)
                        my_s = s0 + (rank - first);
                        got_sample = true;
                    }
                    return true;
                });
                if (got_sample) {  // start the sample
                    const PathArgs pa = reload_path_args();  // (cold: the camera)
                    camera_path(pa, my_col, my_row, pa.sample_offset + my_s, q.p);
                    q.active = true;
                    ++n_paths;
                }
                any_active = any_active || q.active;
            }
        }
        // No live path anywhere in the wave: every slot asked and got nothing, so the pool is
        // used up and the global queue drained (an entry shortage needs live paths to exist).
        if (__ballot(any_active) == 0ull) break;
#ifdef RTIOW_DEBUG_TIMELINE
        if (tl_dry != 0ull) ++tl_tail_iters;
        if (tl_dry != 0ull && tl_live_at_dry == ~0u) {
            tl_live_at_dry = static_cast<uint32_t>(__popcll(__ballot(sl[0].active)) + (kSlots > 1 ? __popcll(__ballot(sl[kSlots - 1].active)) : 0)) + pass_n;
            if (lane == 0u) atomicMax(&a.counters->not_first_dry, ~tl_dry);
        }
#endif
        [[maybe_unused]] const unsigned long long t1 = DBG_STAMP();
#ifndef RTIOW_DEEP_PRIO_FROM
#define RTIOW_DEEP_PRIO_FROM 0  // (A/B: segments from which a path makes its wave's instructions win the SIMD's issue arbitration; 0: off)
#endif
        if constexpr (RTIOW_DEEP_PRIO_FROM != 0) {
            // a wave that carries a long path is on the frame's critical path -- what ends a small frame is the
            // lifetime of its fifty-bounce paths, one bounce per iteration of their wave -- so it issues ahead of its SIMD's other waves
            const bool deep = (sl[0].active && meta_depth(sl[0].meta) >= RTIOW_DEEP_PRIO_FROM) ||
                              (sl[kSlots - 1].active && meta_depth(sl[kSlots - 1].meta) >= RTIOW_DEEP_PRIO_FROM);
            if (__ballot(deep) != 0ull) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        }

        // ---- trace ----------------------------------------------------------
        float best[kSlots];
        int best_i[kSlots];        // slot of the hit in the LDS list, -1: miss
        uint32_t best_o[kSlots];   // its original sphere index
        uint32_t live_paths = 0u;
#pragma unroll
        for (int r = 0; r < kSlots; ++r) live_paths += static_cast<uint32_t>(__popcll(__ballot(sl[r].active)));
#ifdef RTIOW_DEBUG_COUNTERS
        if (dbg_dry_seen) ++dbg_tail_iters;
        const unsigned long long dbg_tr0 = wall_clock64();
#endif
        if constexpr (kSparseLoop) {
            if (exhausted && pass_n == 0u && live_paths <= kSparseParMax) {  // nothing left to draw: the rest of the frame in the sparse loop
                compact_to_slot0(sl, reinterpret_cast<uint32_t*>(lds_results));
                to_sparse_loop = true;
                break;
            }
        }
        if constexpr (!ACCEL) {
            // The end of a frame with the flat list (round 5): once the queues are dry and few paths live, they are gathered in slot 0, so
            // that the shade code runs once an iteration instead of twice (its second pass finds no path and leaves at once).  BASELINE config
            // 2 -- 400 x 225, 0.5 ms -- is a third tail: every wave ends on the fifty bounces of a path inside the glass ball.
            if (exhausted && live_paths <= 32u) compact_to_slot0(sl, reinterpret_cast<uint32_t*>(lds_results));
        }
        if (ACCEL && live_paths <= kSparseMaxAccel) {
            // few paths left: gather them in slot 0 (the shade and refill code below then runs once, not
            // once per slot), then trace them together
            compact_to_slot0(sl, reinterpret_cast<uint32_t*>(lds_results));
            trace_sparse_parallel<kSlots, !SHADE_LDS, FLAT>(lds_spheres, lds_cidx, lds_cbounds, a, lds_items, lds_results, sl, best, best_i,
                                          best_o, n_tests);
            TL_MARK(tl_sparse);
#ifdef RTIOW_DEBUG_TIMELINE
            ++tl_sparse_iters;
            tl_sparse_paths += live_paths;
#endif
            DBG_ADD(dbg_sparse, lane == 0u ? 1u : 0u);
#ifdef RTIOW_DEBUG_COUNTERS
            if (dbg_dry_seen) {
                dbg_sp_ticks += wall_clock64() - dbg_tr0;
                ++dbg_sp_iters;
                dbg_sp_paths += live_paths;
            }
#endif
        } else if (!ACCEL && live_paths <= kSparseMax && live_paths * 3u <= g.n_pad) {
            // (the sphere-parallel trace costs ~60 instructions a path whatever the list, the lock-step pass 22 a sphere for all 128 slots:
            // with a handful of spheres -- BASELINE config 2 has four -- the lock-step pass is the cheaper one even for a single path)
            trace_sparse<kSlots>(lds_spheres, nullptr, g.n_pad, a.n, sl, best, best_i, best_o);
#pragma unroll
            for (int r = 0; r < kSlots; ++r)
                if (sl[r].active) n_tests += a.n;
            DBG_ADD(dbg_sparse, lane == 0u ? 1u : 0u);
        } else if (ACCEL) {
            trace_clustered<kSlots, !SHADE_LDS, FLAT>(lds_spheres, lds_cidx, lds_cbounds, a, lds_items, lds_results, sl, best, best_i, best_o, n_tests,
                                    dbg_slow_trips, dbg_cands, dbg_t_slow);
        } else {
            trace_slots<kSlots>(lds_spheres, g.n_pad, a.n, sl, best, best_i, dbg_slow_trips, dbg_cands, dbg_t_slow);
#pragma unroll
            for (int r = 0; r < kSlots; ++r) {
                best_o[r] = static_cast<uint32_t>(best_i[r]);
                if (sl[r].active) n_tests += a.n;
            }
        }
        DBG_ADD(dbg_iters, lane == 0u ? 1u : 0u);
        [[maybe_unused]] const unsigned long long t2 = DBG_STAMP();

        // ---- shade ----------------------------------------------------------
        // Large scenes keep their shading records in HBM/L2: the records of BOTH slots' hits are asked for here, so that
        // the second slot's round trip (about a microsecond under load) passes while the first slot is shaded.
        [[maybe_unused]] float4 rec0[kSlots], rec1[kSlots];
#pragma unroll
        for (int r = 0; r < kSlots; ++r) {
            rec0[r] = rec1[r] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (!SHADE_LDS && sl[r].active && best_i[r] >= 0) {
                const float4* src = reinterpret_cast<const float4*>(a.shade + best_o[r]);
                rec0[r] = src[0];
                rec1[r] = src[1];
            }
        }
#pragma unroll
        for (int r = 0; r < kSlots; ++r) shade_one(sl[r], best[r], best_i[r], best_o[r], rec0[r], rec1[r]);
        if (static_cast<uint32_t>(__popcll(done_entries)) >= kResolveBatch) resolve_done();
        DBG_ADD(dbg_t_refill, t1 - t0);
        DBG_ADD(dbg_t_trace, t2 - t1);
        DBG_ADD(dbg_t_shade, DBG_STAMP() - t2);
#ifdef RTIOW_DEBUG_COUNTERS
        if (dbg_dry_seen) {
            dbg_tail_cyc[0] += t1 - t0;
            dbg_tail_cyc[1] += t2 - t1;
            dbg_tail_cyc[2] += DBG_STAMP() - t2;
        }
#endif
    }

    // ---- the sparse loop: the end of a wave's frame (see above) ----
    // All paths sit in slot 0, at most kSparseParMax of them; an iteration is the path-parallel trace and one pass of the
    // shade code.
    if constexpr (kSparseLoop) {
        if (to_sparse_loop) {
            for (;;) {
                [[maybe_unused]] const uint32_t live = static_cast<uint32_t>(__popcll(__ballot(sl[0].active)));
                if (live == 0u) break;
#ifdef RTIOW_DEBUG_TIMELINE
                if (tl_dry != 0ull) ++tl_tail_iters;
                TL_MARK(tl_sparse);
                ++tl_sparse_iters;
                tl_sparse_paths += live;
#endif
                float best[kSlots];
                int best_i[kSlots];
                uint32_t best_o[kSlots];
#if RTIOW_SPARSE_BATCHED
                if (SHADE_LDS && live >= kSparseBatchMin && live <= kSparseBatchMax) {  // (small scenes: one level of boxes)
                    trace_sparse_batched<FLAT>(lds_spheres, lds_cidx, lds_cbounds, a, lds_items, lds_results, sl[0], best[0], best_i[0], best_o[0], n_tests);
                } else
#endif
                trace_sparse_parallel<kSlots, !SHADE_LDS, FLAT>(lds_spheres, lds_cidx, lds_cbounds, a, lds_items, lds_results, sl, best, best_i,
                                                                best_o, n_tests);
                float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0;
                if (!SHADE_LDS && sl[0].active && best_i[0] >= 0) {  // (large scenes: the hit's shading record from L2)
                    const float4* src = reinterpret_cast<const float4*>(a.shade + best_o[0]);
                    r0 = src[0];
                    r1 = src[1];
                }
                shade_one(sl[0], best[0], best_i[0], best_o[0], r0, r1);
            }
        }
    }
    resolve_done();  // the wave's last pixels

#ifdef RTIOW_DEBUG_TIMELINE
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t d2 = __shfl_down(tl_deepest, off);
        tl_deepest = d2 > tl_deepest ? d2 : tl_deepest;
        const unsigned long long e2 = __shfl_down(tl_deep_end, off);
        tl_deep_end = e2 > tl_deep_end ? e2 : tl_deep_end;
    }
    if (lane == 0u) {
        const unsigned long long t0w = ~__hip_atomic_load(&a.counters->not_t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long stamps[3] = {tl_dry, tl_sparse, static_cast<unsigned long long>(wall_clock64())};
        {
            const uint32_t wv = blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u;
            if (wv < 8192u) {
                unsigned int* rec = a.counters->tl_wave[wv];
                auto rel = [&](unsigned long long t) { return t == 0ull ? 0u : static_cast<unsigned int>(t > t0w ? t - t0w : 1ull); };
                rec[0] = rel(tl_dry); rec[1] = rel(tl_sparse); rec[2] = rel(stamps[2]);
                rec[3] = tl_tail_iters; rec[4] = tl_sparse_iters; rec[5] = (tl_live_at_dry & 0xFFu) | (tl_sparse_paths << 8); rec[6] = tl_deepest; rec[7] = rel(tl_deep_end);
            }
        }
        for (int k = 0; k < 3; ++k) {
            if (stamps[k] == 0ull) continue;
            const unsigned long long b = (stamps[k] > t0w ? stamps[k] - t0w : 0ull) / 5000ull;  // 50 us bins
            atomicAdd(&a.counters->tl_hist[k][b > 63ull ? 63ull : b], 1u);
        }
        atomicMax(&a.counters->tl_tail_iters_max, tl_tail_iters);
        // start of the frame: entry -> scene staged and barrier passed -> first pool fetched -> first hand_out done -> first camera rays made
        atomicAdd(&a.counters->tl_start_sum[0], tl_start - tl_entry);
        atomicMax(&a.counters->tl_start_max[0], static_cast<unsigned int>(tl_start - tl_entry));
        if (tl_first_rays != 0ull) {
            atomicAdd(&a.counters->tl_start_sum[1], tl_first_rays - tl_start);
            atomicMax(&a.counters->tl_start_max[1], static_cast<unsigned int>(tl_first_rays - tl_start));
            atomicAdd(&a.counters->tl_start_sum[2], 1ull);
            atomicAdd(&a.counters->tl_start_sum[3], tl_first_pool - tl_start);
            atomicAdd(&a.counters->tl_start_sum[4], tl_first_out - tl_start);
        }
        atomicAdd(&a.counters->tl_starved_sum, static_cast<unsigned long long>(tl_starved));
        atomicMax(&a.counters->tl_starved_max, tl_starved);
        if (tl_live_at_dry != ~0u) {
            atomicMax(&a.counters->tl_live_at_dry_max, tl_live_at_dry);
            atomicAdd(&a.counters->tl_live_at_dry_hist[tl_live_at_dry / 16u > 8u ? 8u : tl_live_at_dry / 16u], 1u);
            const unsigned long long first_dry = ~__hip_atomic_load(&a.counters->not_first_dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tl_dry > first_dry + 10000ull)
                atomicAdd(&a.counters->tl_late_dry_live_hist[tl_live_at_dry / 16u > 8u ? 8u : tl_live_at_dry / 16u], 1u);
        }
        atomicAdd(&a.counters->tl_tail_iters_sum, static_cast<unsigned long long>(tl_tail_iters));
        if (tl_sparse != 0ull) {
            atomicAdd(&a.counters->tl_sparse_iters_sum, static_cast<unsigned long long>(tl_sparse_iters));
            atomicAdd(&a.counters->tl_sparse_paths_sum, static_cast<unsigned long long>(tl_sparse_paths));
            atomicAdd(&a.counters->tl_sparse_ticks_sum, stamps[2] - tl_sparse);
        }
    }
#endif
    // One counter update per WORKGROUP: the waves sum in LDS as they leave, the last one out adds the sums to the global
    // counters (a global atomic is a 64-byte memory-side request: three per wave were 0.6 MB per frame).  A wave's sums
    // reach LDS before its arrival tick (its LDS operations are performed in order), so the last arrival sees them all.
    const PathArgs ea = reload_path_args();  // (cold: the counter blocks)
    bool last_group = false;
    // the lanes' counts summed on the DPP network, totals in lane 63 (rounds 1-4: __shfl_down, whose lane number the compiler took from the
    // kernel's first block and kept -- spilled, in the COMPACT variant -- to this last one); a wave's tests in two halves: 64 x 2^32 does not fit
    if constexpr (!kWaveCounts) {  // (the large-scene variants start every path in a pass)
        n_paths = wave_inclusive_sum(n_paths);
        n_segments = wave_inclusive_sum(n_segments);
    }
    const unsigned long long tests64 = (static_cast<unsigned long long>(wave_inclusive_sum(n_tests >> 16)) << 16) + wave_inclusive_sum(n_tests & 0xFFFFu);
    if (lane == 63u) {
        atomicAdd(&wg_sums[0], static_cast<unsigned long long>(kWaveCounts ? wave_paths : n_paths));
        atomicAdd(&wg_sums[1], static_cast<unsigned long long>(kWaveCounts ? wave_segments : n_segments));
        atomicAdd(&wg_sums[2], tests64);
        // (release / acquire at workgroup scope on the arrival tick, and atomic reads of the sums: the ordering the last
        // wave relies on is in the code, not in how the LDS happens to execute a wave's operations)
        if (__hip_atomic_fetch_add(wg_left_p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u == waves_in_group) {
            atomicAdd(&ea.counters->paths, __hip_atomic_load(&wg_sums[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            atomicAdd(&ea.counters->segments, __hip_atomic_load(&wg_sums[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            atomicAdd(&ea.counters->tests, __hip_atomic_load(&wg_sums[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            // the last workgroup out zeroes the counter block of the NEXT frame (queue heads and all): the frames of a
            // loop then follow each other without a memset in between (with the stats copy moved to rtGetStats, the gap
            // between two path kernels went from 34 to ~10 us)
            last_group = __hip_atomic_fetch_add(&ea.counters->wg_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x;
        }
    }
    if (blockIdx.x == 0u && threadIdx.x == 0u) {  // (its own stamps: written by this very lane)
        ea.counters->clk_cycles = __builtin_readcyclecounter() - clk_start[0];  // (rtGetStats divides)
        ea.counters->clk_ticks = wall_clock64() - clk_start[1];
    }
    last_group = __builtin_amdgcn_readlane(static_cast<uint32_t>(last_group), 63) != 0u;  // (lane 63 holds the sums and the tick)
    if (last_group && ea.next_counters != nullptr) {
        uint32_t* words = reinterpret_cast<uint32_t*>(ea.next_counters);
        for (uint32_t k = lane; k < sizeof(Counters) / 4u; k += 64u) words[k] = 0u;
    }
#ifdef RTIOW_DEBUG_COUNTERS
    // debug[0] wave-level slow-loop trips, [1] lane-level candidates, [2] wave iterations
    atomicAdd(&a.counters->debug[0], static_cast<unsigned long long>(dbg_slow_trips));
    atomicAdd(&a.counters->debug[1], static_cast<unsigned long long>(dbg_cands));
    atomicAdd(&a.counters->debug[2], static_cast<unsigned long long>(dbg_iters));
    // sparse iterations are reported in the high half of debug[2]
    atomicAdd(&a.counters->debug[2], static_cast<unsigned long long>(dbg_sparse) << 32);
    if (lane == 0u) {  // per-wave cycle shares: [3] refill [4] trace (incl. slow) [5] slow [6] shade
        atomicAdd(&a.counters->debug[3], dbg_t_refill);
        atomicAdd(&a.counters->debug[4], dbg_t_trace);
        atomicAdd(&a.counters->debug[5], dbg_t_slow);
        atomicAdd(&a.counters->debug[6], dbg_t_shade);
        for (int k = 0; k < 10; ++k) atomicAdd(&a.counters->pass_stats[k], dbg_pass[k]);
        {
            const unsigned long long t0w = ~__hip_atomic_load(&a.counters->not_t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long b = (wall_clock64() - t0w) / 12500ull;
            atomicAdd(&a.counters->hist_end[b > 31ull ? 31ull : b], 1u);
            if (dbg_dry_seen) {
                atomicAdd(&a.counters->tail_iters, static_cast<unsigned long long>(dbg_tail_iters));
                atomicAdd(&a.counters->tail_ticks, wall_clock64() - dbg_dry_at);
                atomicAdd(&a.counters->tail_sparse_iters, static_cast<unsigned long long>(dbg_sp_iters));
                atomicAdd(&a.counters->tail_sparse_ticks, dbg_sp_ticks);
                atomicAdd(&a.counters->tail_sparse_paths, static_cast<unsigned long long>(dbg_sp_paths));
                for (int k = 0; k < 3; ++k) atomicAdd(&a.counters->tail_cyc[k], dbg_tail_cyc[k]);
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {  // clock = shader cycles per 100 MHz tick
            const unsigned long long dc = DBG_STAMP() - dbg_c0, dw = wall_clock64() - dbg_w0;
            a.counters->debug[7] = dw ? dc * 100ull / dw : 0ull;  // MHz
        }
    }
#endif
}

#ifndef RTIOW_TU_PART
// ============================================================================
// arithmetic conformance probe (tests/test_gpu_parity.py::test_arith_bit_exact)
// ============================================================================
__global__ void arith_kernel(uint32_t op, const float* a, const float* b, const float* c,
                             float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (op) {
        case 0: r = fma_(a[i], b[i], c[i]); break;
        case 1: r = a[i] / b[i]; break;
        case 2: r = __builtin_sqrtf(a[i]); break;
        case 3: r = a[i] * b[i]; break;
        case 4: r = a[i] + b[i]; break;
        case 5: {
            Pcg rng(__float_as_uint(a[i]), __float_as_uint(c[i]), 7u);
            for (int k = 0; k < 4; ++k) r = rng.uniform();
            break;
        }
        case 6: r = static_cast<float>(to_fixed(a[i]) + to_fixed(b[i])); break;
        case 7:
            r = static_cast<float>((static_cast<unsigned long long>(__float_as_uint(a[i])) << 32) |
                                   __float_as_uint(b[i]));
            break;
        case 8: r = lean_sqrt(a[i]); break;      // the PATH and CH kernels' lean forms themselves (psqrt / pdiv; ch_pixel<LEAN>):
        case 9: r = lean_div(a[i], b[i]); break; // against sqrtf and / on operands inside their preconditions
        // the two-phase CH pixels piece by piece (results are bit patterns; bit 31 set: the second phase ran)
        case 10: r = ch_rsq(a[i]); break;                                                              // the instruction itself: one ulp?
        case 11: r = __uint_as_float(ch_sky_two_phase(reinterpret_cast<const uint4*>(ch_sky_table_words), a[i], b[i])); break;  // (dy, qa)
        case 12: r = __uint_as_float(ch_sky_colour<true>(lean_div(a[i], lean_sqrt(b[i])))); break;    // the exact sky of (dy, qa)
        case 13: r = __uint_as_float(ch_sky_colour<true>(a[i])); break;                                // F(unit_y)
        case 14: r = __uint_as_float(ch_normal_two_phase(mk(a[i], b[i], c[i]))); break;                // the normal colour of v
        case 15: { const f3 v = mk(a[i], b[i], c[i]); r = __uint_as_float(ch_normal_colour<true>(v, lean_sqrt(gdot(v, v)))); break; }
        case 17: r = newton_sqrt(a[i]); break;  // the six-instruction square root of the PATH kernels and the two-phase CH pixels
        case 18: r = newton_rcp(a[i]); break;   // the three-instruction reciprocal of unit3_scattered
        case 16: { bool second; const uint32_t col = ch_sky_phase1(reinterpret_cast<const uint4*>(ch_sky_table_words), a[i], second); r = __uint_as_float(second ? kChPhase2Flag : col); break; }  // the table alone
        default: break;
    }
    out[i] = r;
}

// Every float unit_y in [lo, hi] at which the sky colour F (ch_sky_colour<true>, the shader's arithmetic) differs from F of the
// float before it: the step list tools/gen_ch_sky_table.py derives on the host from its own restatement, here from the device code
// itself on all two billion floats (tests/test_gpu_parity.py compares the two).  key: floats in ascending order, -k for negative ones.
DI float ch_float_of_key(long long k) { return __uint_as_float(k < 0 ? static_cast<uint32_t>(-k) | 0x80000000u : static_cast<uint32_t>(k)); }
__global__ __launch_bounds__(256) void ch_sky_steps_kernel(long long key_lo, unsigned long long n, RtChSkyStep* out, uint32_t cap, uint32_t* count) {
    const unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * 256ull + threadIdx.x;
    if (i >= n) return;
    const long long key = key_lo + static_cast<long long>(i);
    const float y = ch_float_of_key(key);
    const uint32_t before = ch_sky_colour<true>(ch_float_of_key(key - 1)), after = ch_sky_colour<true>(y);
    if (before != after) {
        const uint32_t at = atomicAdd(count, 1u);
        if (at < cap) out[at] = RtChSkyStep{y, before, after};
    }
}
// newton_sqrt against sqrtf (fn 0) or newton_rcp against 1.0f / x (fn 1) on every float of [lo, hi] (bit patterns of positive floats,
// ascending): the number of floats on which the two differ, and the first few of them.  rtSelfTestUnaryScan; the proof of the two short
// forms is this kernel's zero over [2^-96, 2^128) and [2^-64, 2^64).
__global__ __launch_bounds__(256) void unary_scan_kernel(uint32_t fn, uint32_t lo_bits, unsigned long long n, unsigned long long* bad, uint32_t* first,
                                                         uint32_t cap) {
    const unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * 256ull + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float(lo_bits + static_cast<uint32_t>(i));
    const float got = fn == 0u ? newton_sqrt(x) : newton_rcp(x), want = fn == 0u ? __builtin_sqrtf(x) : 1.0f / x;
    if (__float_as_uint(got) != __float_as_uint(want)) {
        const unsigned long long at = atomicAdd(bad, 1ull);
        if (at < cap) first[at] = __float_as_uint(x);
    }
}

// ============================================================================
// chunk order for the next frame (cost-ordered dequeue): one workgroup, counting sort on 1024 cost classes
// ============================================================================
constexpr uint32_t kOrderBins = 1024;
// Classes on a fixed scale, so that one look at the costs suffices for the histogram: a chunk's cost over the samples
// of one of its pixels is 32 x its mean segments per sample; class = 1023 - (that - 32), clamped: steps of 1/32 segment
// from a mean of 1 (sky: every path one segment) to a mean of 33, anything dearer in class 0.  (The first version
// looked for the largest cost first: three passes of thirty dependent-latency loads per thread, 46 us per frame of the
// cover scene, serial behind the path kernel.  Loads are issued eight at a time now.)
DI uint32_t cost_class(unsigned long long cost, float inv_spp) {
    const float units = static_cast<float>(cost) * inv_spp - 32.0f;
    const float c = units > 0.0f ? (units < static_cast<float>(kOrderBins - 1u) ? units : static_cast<float>(kOrderBins - 1u)) : 0.0f;
    return (kOrderBins - 1u) - static_cast<uint32_t>(c);
}
__global__ __launch_bounds__(1024) void order_chunks_kernel(unsigned long long* cost, uint32_t* order, uint32_t n, float inv_spp) {
    __shared__ uint32_t first[kOrderBins];  // histogram, then the first place of each class
    __shared__ uint32_t any;
    const uint32_t t = threadIdx.x;
    first[t] = 0u;
    if (t == 0u) any = 0u;
    __syncthreads();
    constexpr uint32_t kBatch = 8;
    bool seen = false;
    for (uint32_t c0 = t; c0 < n; c0 += kBatch * kOrderBins) {
        unsigned long long v[kBatch];
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) v[u] = c0 + u * kOrderBins < n ? cost[c0 + u * kOrderBins] : 0ull;
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) {
            // (a third of the cover frame is sky, all of it in the last class: those lanes are counted with a
            // ballot instead of ten thousand LDS atomics on one word)
            const bool in = c0 + u * kOrderBins < n;
            const uint32_t k = in ? cost_class(v[u], inv_spp) : 0u;
            const bool last = in && k == kOrderBins - 1u;
            const unsigned long long m = __ballot(last);
            if (in && !last) atomicAdd(&first[k], 1u);
            if ((t & 63u) == 0u && m != 0ull) atomicAdd(&first[kOrderBins - 1u], static_cast<uint32_t>(__popcll(m)));
            seen = seen || (in && v[u] != 0ull);
        }
    }
    if (seen) any = 1u;
    __syncthreads();
    // exclusive prefix sum over the classes (thread t owns class t): Hillis-Steele in LDS
    uint32_t incl = first[t];
    const uint32_t own = incl;
    for (uint32_t off = 1u; off < kOrderBins; off <<= 1u) {
        __syncthreads();
        first[t] = incl;
        __syncthreads();
        if (t >= off) incl += first[t - off];
    }
    __syncthreads();
    first[t] = incl - own;
    __syncthreads();
    // place s of the sequence belongs to queue s % 8, which reads it as its (s / 8)-th chunk: stored queue by queue
    auto slot_of = [&](uint32_t seq) { return chunk_order_slot(seq, n); };  // (< chunk_order_words(n): rtRender allocates that many)
    if (any == 0u) {  // nothing measured: natural order
        for (uint32_t c = t; c < n; c += kOrderBins) order[slot_of(c)] = c;
        return;
    }
    for (uint32_t c0 = t; c0 < n; c0 += kBatch * kOrderBins) {
        unsigned long long v[kBatch];
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) v[u] = c0 + u * kOrderBins < n ? cost[c0 + u * kOrderBins] : 0ull;
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) {
            const bool in = c0 + u * kOrderBins < n;
            const uint32_t k = in ? cost_class(v[u], inv_spp) : 0u;
            const bool last = in && k == kOrderBins - 1u;
            const unsigned long long m = __ballot(last);
            uint32_t base = 0u;
            if ((t & 63u) == 0u && m != 0ull) base = atomicAdd(&first[kOrderBins - 1u], static_cast<uint32_t>(__popcll(m)));
            base = __builtin_amdgcn_readfirstlane(base);
            if (in) {
                const uint32_t place = last ? base + lane_rank(m) : atomicAdd(&first[k], 1u);
                order[slot_of(place)] = c0 + u * kOrderBins;
                cost[c0 + u * kOrderBins] = 0ull;
            }
        }
    }
}

#endif  // RTIOW_TU_PART
}  // namespace

// The small-scene clustered kernels -- the default kernel of the headline frame -- are compiled in a second pass over this
// file (Makefile: -DRTIOW_TU_SMALL_CLUSTERED, object rtiow_kernels_small.o) with a scheduler and a register budget of their
// own.  Round 2: -mllvm -amdgpu-sched-strategy=iterative-ilp, 159 registers, three waves per SIMD (cover frame 8.64 -> 8.40
// ms against the default scheduler).  Round 3: iterative-minreg brings the flat-axis variant to 136 registers; held to 128
// (-DRTIOW_SMALL_WAVES_PER_EU=4) it spilled 13 of them (28 by the end of the round) and runs FOUR waves per SIMD -- launch_path
// then finds two groups of 512 threads per CU: cover frame 7.95 -> 7.19 ms, one eighth of it 1.33 -> 1.26 (interleaved A/B,
// tools/ab_bench.py; iterative-minreg at three waves: 8.48).  Round 4: with Slot's bookkeeping in one register and one base for the per-wave LDS areas, none;
// round 5: ONE group of 1024 threads per CU (one copy of the scene: RTIOW_PASS_KEEP), the default scheduler, 113 / 120 registers.  The large-scene variants spill 115 registers at 128 and have no LDS for sixteen
// waves' buffers, the flat-list kernels (95 registers) gain nothing: they stay with the default scheduler.
// Round 4: the large-scene variants get a pass of their own too (-DRTIOW_TU_LARGE_CLUSTERED, rtiow_kernels_large.o) under iterative-ilp: at 148
// registers (three waves per SIMD allow 168) it spills nothing and the 4099-sphere scene renders 1.8 % faster than under the default
// scheduler (max-ilp: +1 %, iterative-minreg: +5 %; profiles/r04_scheduler_ab.txt); the flat-list kernels keep the default one.
#if defined(RTIOW_TU_SMALL_CLUSTERED)
PersistentKernelFn small_clustered_kernel(bool flat) {
    return flat ? path_persistent_kernel<true, true, true> : path_persistent_kernel<true, true, false>;
}
#elif defined(RTIOW_TU_LARGE_CLUSTERED)
PersistentKernelFn large_clustered_kernel(bool flat) {
    return flat ? path_persistent_kernel<false, true, true> : path_persistent_kernel<false, true, false>;
}
#else

hipError_t launch_order_chunks(unsigned long long* cost, uint32_t* order, uint32_t n_chunks, uint32_t spp, hipStream_t stream) {
    hipLaunchKernelGGL(order_chunks_kernel, dim3(1), dim3(kOrderBins), 0, stream, cost, order, n_chunks,
                       1.0f / static_cast<float>(spp ? spp : 1u));
    return hipGetLastError();
}

hipError_t launch_ch(const ChArgs& args, hipStream_t stream) {
    ChArgs a = args;
    // one row or one column (u or v is 0/0), or the tiled form asked for: one lane per pixel, as the reference dispatches it
    if (a.width < 2u || a.height < 2u || a.tiles_form != 0u) {
        const uint32_t tiles = ((a.width + 15u) / 16u) * ((a.height + 15u) / 16u);
        hipLaunchKernelGGL(ch_kernel_tiles, dim3(tiles), dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    // the lean square roots and quotients (lean_sqrt, lean_div) for a camera of moderate proportions, hipcc's full forms otherwise
    auto moderate = [](float v) { return std::fabs(v) >= 0x1p-20f && std::fabs(v) <= 0x1p20f; };
    const bool lean = moderate(a.ubo.viewportWidth) && moderate(a.ubo.viewportHeight) && moderate(a.ubo.focalLength) &&
                      !debug_knob("RTIOW_DEBUG_CH_FULL");  // (the variable: A/B and parity tests)
    // ... and of those the two-phase pixels, whose kernel also takes u = column / (imageWidth - 1) and v by the lean quotient: divisors in
    // [1, 2^24] (RTIOW_DEBUG_CH_LEAN, knobs build: the exact lean kernel, for A/B and parity tests)
    auto countable = [](float v) { return v - 1.0f >= 1.0f && v - 1.0f <= 0x1p24f; };
    const bool two_phase = lean && countable(a.ubo.imageWidth) && countable(a.ubo.imageHeight) && !debug_knob("RTIOW_DEBUG_CH_LEAN");
    const uint32_t tiles_x = (a.width + kChTileCols - 1u) / kChTileCols;
    uint32_t rpw = 1u;
    if (two_phase) {
        // Two-phase kernel: a wave's rows are its serial path (a row of sphere pixels is ~300 instructions) and the chip holds 2048
        // workgroups at a time, so rows are added to a wave only while that leaves 4096 workgroups -- 8192 for the step to 16 rows.
        // Measured (tools/ch_bandwidth.py, ms): 4096^2 with 2 / 4 / 8 rows per wave 0.0261 / 0.0247 / 0.0282; 8192^2 with 4 / 8 / 16:
        // 0.0666 / 0.0655 / 0.074; 16384^2 at 16: 0.219.  The reference's 800 x 608: 608 workgroups of 4 rows.
        while (rpw < 16u && static_cast<unsigned long long>(tiles_x) * ((a.height + 8u * rpw - 1u) / (8u * rpw)) >= (rpw == 8u ? 8192ull : 4096ull))
            rpw *= 2u;
    } else {
        // rows a wave renders: enough workgroups to fill the chip eight times over first (the reference's 800x608 frame is
        // launch-bound: 4 rows per workgroup, 608 workgroups), then up to 16 so that the per-column values are reused
        while (rpw < 16u && static_cast<unsigned long long>(tiles_x) * ((a.height + 8u * rpw - 1u) / (8u * rpw)) >= 2048ull) rpw *= 2u;
    }
    a.rows_per_wave = rpw;
    a.vector_store = (reinterpret_cast<uintptr_t>(a.dst) % 16u == 0u && a.dst_stride % 4u == 0u) ? 1u : 0u;
    a.row_blocks = (a.height + 4u * rpw - 1u) / (4u * rpw);
    const uint32_t blocks = tiles_x * a.row_blocks;
    // the two-phase kernel's order of row blocks: a stride near the golden section of their number, coprime with it (1: raster order)
    a.row_block_stride = 1u;
    if (a.row_blocks > 4u && !debug_knob("RTIOW_DEBUG_CH_RASTER")) {
        auto gcd = [](uint32_t x, uint32_t y) { while (y) { const uint32_t t = x % y; x = y; y = t; } return x; };
        uint32_t stride = static_cast<uint32_t>(a.row_blocks * 0.6180339887) | 1u;
        while (gcd(stride, a.row_blocks) != 1u) stride += 2u;
        a.row_block_stride = stride % a.row_blocks;
    }
    if (two_phase) hipLaunchKernelGGL(ch_kernel_rows<kChTwoPhase>, dim3(blocks), dim3(256), 0, stream, a);
    else if (lean) hipLaunchKernelGGL(ch_kernel_rows<kChLean>, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(ch_kernel_rows<kChFull>, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

static bool use_persistent(uint32_t kernel) { return kernel != KERNEL_PIXEL; }
constexpr uint32_t kClusteredFrom = 64;  // spheres; below, boxes + one cluster cost more than the flat scan


// Host side of the primary pass's cone cull (see "The primary pass"): the bounds and margins cone_of_span / cone_reaches /
// cone_reaches_sphere work with, in double.  False: no cull for this camera (degenerate image, non-finite camera, or a
// lens that leaves the range the cluster boxes were inflated for).
static bool cone_setup(const PathArgs& a, PersistArgs& g) {
    const RtCamera& c = a.cam;
    auto norm2 = [](const float* v) { return double(v[0]) * v[0] + double(v[1]) * v[1] + double(v[2]) * v[2]; };
    const double uu = norm2(c.u), vv = norm2(c.v);
    const double uv = double(c.u[0]) * c.v[0] + double(c.u[1]) * c.v[1] + double(c.u[2]) * c.v[2];
    // |x u + y v| <= sigma_max |(x, y)|, and the lens sample has |(x, y)| <= lens_radius
    const double sigma = std::sqrt(0.5 * (uu + vv) + 0.5 * std::sqrt((uu - vv) * (uu - vv) + 4.0 * uv * uv));
    const double lens = c.lens_radius > 0.0f ? double(c.lens_radius) * sigma * (1.0 + 1.0 / 256.0) : 0.0;
    g.lens_rho = static_cast<float>(lens);
    g.h_len = static_cast<float>(std::sqrt(norm2(c.horizontal)) * (1.0 + 1.0 / 64.0));
    g.v_len = static_cast<float>(std::sqrt(norm2(c.vertical)) * (1.0 + 1.0 / 64.0));
    double cmax = 0.0, omax = 0.0, dist2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        cmax = std::max(cmax, std::fabs(double(a.ccenter[k])));
        omax = std::max(omax, std::fabs(double(c.origin[k])));
        dist2 += (double(c.origin[k]) - a.ccenter[k]) * (double(c.origin[k]) - a.ccenter[k]);
    }
    const double rmax = std::sqrt(double(a.crmax2));
    g.abs_margin = static_cast<float>((cmax + omax + 1.5 * rmax) / 65536.0);
    const bool finite = std::isfinite(lens) && std::isfinite(g.h_len) && std::isfinite(g.v_len) && std::isfinite(g.abs_margin) &&
                        std::isfinite(dist2);
    // (the boxes hold for ray origins within rmax of the scene's centre; rtRender re-boxes for a camera beyond 0.95 rmax)
    const bool in_range = std::sqrt(dist2) + lens <= 0.97 * rmax;
    return finite && in_range && a.width >= 2u && a.height >= 2u;
}

// The cone cull on the host, for the CPU test of its conservativeness (tests/test_host_logic.py): the very functions the
// primary pass runs, for the span of pixels pix_lo..pix_hi (one row) of a width x height image -- which of the spheres
// {cx, cy, cz, radius} and of the boxes {centre xyz, half extent xyz} the cones of the span may reach.  Returns 0 when
// the cull is off for this camera (everything is then reached).
int cone_selftest_host(const RtCamera& cam, uint32_t width, uint32_t height, uint32_t pix_lo, uint32_t pix_hi,
                       const float* range_center, float range_rmax, const RtSphere* spheres, uint32_t n_spheres,
                       const float* boxes, uint32_t n_boxes, uint8_t* sphere_reach, uint8_t* box_reach) {
    PathArgs a{};
    a.cam = cam;
    a.width = width;
    a.height = height;
    a.inv_wm1 = 1.0f / static_cast<float>(width - 1u);
    a.inv_hm1 = 1.0f / static_cast<float>(height - 1u);
    a.row_block = 1u;
    a.tile_count = 1u;
    for (int k = 0; k < 3; ++k) a.ccenter[k] = range_center[k];
    a.crmax2 = range_rmax * range_rmax;
    PersistArgs g{};
    const bool cull = cone_setup(a, g);
    const ConeAxis c = cull ? cone_of_span(a, g, pix_lo % width, pix_lo % width + (pix_hi - pix_lo), pix_lo / width) : ConeAxis{};
    for (uint32_t i = 0; i < n_spheres; ++i) {
        const float4 s = make_float4(spheres[i].cx, spheres[i].cy, spheres[i].cz, spheres[i].radius * spheres[i].radius);
        sphere_reach[i] = !cull || cone_reaches_sphere(a, g, c, s) ? 1u : 0u;
    }
    for (uint32_t i = 0; i < n_boxes; ++i) {
        const float4 mid = make_float4(boxes[6u * i], boxes[6u * i + 1u], boxes[6u * i + 2u], 0.0f);
        const float4 half = make_float4(boxes[6u * i + 3u], boxes[6u * i + 4u], boxes[6u * i + 5u], 0.0f);
        box_reach[i] = !cull || cone_reaches(a, g, c, mid, half) ? 1u : 0u;
    }
    return cull ? 1 : 0;
}

hipError_t launch_path(const PathArgs& args, uint32_t kernel, uint32_t max_take, int num_cus,
                       hipStream_t stream, uint32_t* resolved) {
    PathArgs a = args;
    if (!use_persistent(kernel)) {
        *resolved = KERNEL_PIXEL;
        const uint32_t tiles = ((a.width + 15u) / 16u) * ((a.local_rows + 15u) / 16u);
        const size_t lds = static_cast<size_t>(a.n) * sizeof(float4);
        if (lds > 48u * 1024u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(path_pixel_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(path_pixel_kernel, dim3(tiles), dim3(256), lds, stream, a);
        return hipGetLastError();
    }
    // default: the clustered list from kClusteredFrom spheres on (cover scene: 2.1x faster than the
    // flat list; frames are byte-identical either way), the flat list for the handful-of-spheres scenes
    bool accel = kernel == KERNEL_CLUSTERED || kernel == KERNEL_CLUSTERED_PASS || (kernel == KERNEL_DEFAULT && a.n >= kClusteredFrom);
    // The clustered list must fit the LDS beside four waves' buffers.  The very largest scenes give up
    // the super-cluster level first (its boxes and second work list), then fall back to the flat list.
    // the flat axis (rtiow_clusters.cpp): RTIOW_DEBUG_FLAT=0 tests the whole boxes all the same (A/B and parity tests;
    // RTIOW_DEBUG_FLAT=1, read by rtSetScene, makes every scene flat along its best axis)
    if (!accel || (debug_knob("RTIOW_DEBUG_FLAT") && atoi(debug_knob("RTIOW_DEBUG_FLAT")) == 0)) a.flat_axis = 3u;
    const bool flat = a.flat_axis < 3u;
    const uint32_t box_bytes = flat ? 16u : 32u;  // (LDS per box: without the flat axis one float4 instead of two)
    if (accel) {
        const int levels = clustered_levels_that_fit(a.n_cslots, a.n_clusters, a.n_super, flat);
        if (levels < 2) a.n_super = 0u;
        if (levels < 1) accel = false;
    }
    const uint32_t item_bytes = wave_item_bytes(a.n_super != 0u);
    *resolved = accel ? KERNEL_CLUSTERED : KERNEL_PERSISTENT;
    PersistArgs g{};
    // slots of the LDS sphere list: the flat list padded to whole candidate words, or the clustered one
    g.n_pad = accel ? a.n_cslots : (a.n + 3u) / 4u * 4u;
    (void)max_take;  // scheduling is per sample now; the hint is accepted and ignored
    g.total_pix = a.local_rows * a.width;
    // tests a segment costs, roughly: the whole list, or large spheres + boxes (14 of 11 instructions)
    // + 1.4 clusters of 16 at half the lane efficiency
    const uint32_t seg_cost = accel ? a.n_large + (a.n_super ? a.n_super + 24u : a.n_clusters) * 14u / 11u + 44u : a.n;
    uint32_t pool_samples = kPoolWork / (seg_cost < 1u ? 1u : seg_cost);
    pool_samples = pool_samples < 256u ? 256u : (pool_samples > 4096u ? 4096u : pool_samples);
    g.pool_pix = pool_samples / a.spp;  // a few pixels per pool; one pixel when spp is large
    g.pool_pix = g.pool_pix < 1u ? 1u : (g.pool_pix > 256u ? 256u : g.pool_pix);
    if (const char* v = debug_knob("RTIOW_DEBUG_POOL_PIX")) g.pool_pix = strtoul(v, nullptr, 10);  // tuning only
    // LDS per workgroup: the sphere list (16 B per slot; clustered: + 4 B per slot of indices and
    // 32 B per cluster box); while the scene is small, the shading records too (32 B each);
    // 2 KiB of pixel accumulator entries and 0.5 KiB of line buffers per wave (clustered: + 2-3 KiB of work lists
    // and result keys).
    const size_t lds_geo = static_cast<size_t>(g.n_pad) * sizeof(float4) +
                           (accel ? static_cast<size_t>(g.n_pad) * sizeof(SlotIndex) + static_cast<size_t>(a.n_clusters + a.n_super) * box_bytes : 0u);
    // (a scene with super-clusters -- more than kSuperFrom clusters, i.e. more than 512 small spheres: 30 KB at least -- is never a
    // small one: the small-scene kernels are compiled without that level)
    const bool shade_lds = a.n_super == 0u && lds_geo + static_cast<size_t>(a.n) * sizeof(ShadeRec) <= 28u * 1024u &&
                           !debug_knob("RTIOW_DEBUG_NO_SHADE_LDS");  // (tuning only)
    const size_t lds_scene = lds_geo + (shade_lds ? static_cast<size_t>(a.n) * sizeof(ShadeRec) : 0u);
    // the clustered kernels' primary pass keeps up to pass_keep camera paths per wave in LDS records of their own; a
    // large scene with no room for them (C5: one 768-thread group beside 92 KB of list) does without -- its passes then
    // make no more paths than there are idle slots, and the records sit in the two-level work-list area alone
    // (flat list: 2 KiB of scratch for gathering a wave's last paths in one slot, compact_to_slot0)
    auto wave_bytes_of = [&](uint32_t keep, bool compact) {
        const uint32_t acc_e = compact ? kAccEntriesCompact : kAccEntries, line_b = compact ? kLineBufsCompact : kLineBufs;
        return acc_e * kAccWords * 8u + acc_e * 4u + line_b * (kChunkPix + kLineMetaWords) * 4u + (accel ? item_bytes + keep * kPassRecBytes : 32u * 64u);
    };
    void (*kernel_fn)(PathArgs, PersistArgs) =
        accel ? (shade_lds ? small_clustered_kernel(flat) : large_clustered_kernel(flat))
              : (shade_lds ? path_persistent_kernel<true, false> : path_persistent_kernel<false, false>);
    // (large scenes: the same kernel with the compact per-wave area, four waves per SIMD -- taken when that keeps more waves on a CU)
    void (*compact_fn)(PathArgs, PersistArgs) = nullptr;
    if (accel && !shade_lds && !debug_knob("RTIOW_DEBUG_NO_COMPACT"))
        compact_fn = flat ? path_persistent_kernel<false, true, true, true> : path_persistent_kernel<false, true, false, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsPerCu));
    if (e != hipSuccess) return e;
    if (compact_fn != nullptr) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(compact_fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsPerCu));
        if (e != hipSuccess) return e;
    }
    // Workgroup size: the one that keeps the most waves on a CU (the waves of a group share one copy of
    // the scene, so small scenes do best with 256-thread groups and large ones with 1024); ties go to
    // the smaller group.  RTIOW_DEBUG_THREADS pins it (tuning only).
    uint32_t threads = 0u;
    int per_cu = 0;
    size_t lds = 0u;
    const uint32_t pinned = debug_knob("RTIOW_DEBUG_THREADS") ? strtoul(debug_knob("RTIOW_DEBUG_THREADS"), nullptr, 10) : 0u;
    // (the small-scene variant of the clustered kernel is compiled for groups of at most 512: with the bound at
    // 768 the same source came out 3 % slower on the cover frame)
    const uint32_t t_max = accel ? (shade_lds ? static_cast<uint32_t>(RTIOW_SMALL_MAX_THREADS) : static_cast<uint32_t>(kAccelMaxThreads)) : 1024u;
    uint32_t keep_env = kPassKeep;
    if (const char* v = debug_knob("RTIOW_DEBUG_PASS_KEEP")) keep_env = strtoul(v, nullptr, 10) ? kPassKeep : 0u;  // tuning only
    void (*chosen_fn)(PathArgs, PersistArgs) = kernel_fn;
    // (The search below asks the occupancy API up to 72 times: its result is remembered per thread for the last few (kernel, scene size)
    // pairs -- a frame loop launches the same configuration every frame, and rtRender's host cost is part of a small frame's rate.)
    struct LaunchChoice {
        const void *fn = nullptr, *compact = nullptr;
        size_t lds_scene = 0;
        uint32_t item_bytes = 0, keep_env = 0, pinned = 0, t_max = 0;
        uint32_t threads = 0, pass_keep = 0, wave_bytes = 0;
        int per_cu = 0;
        size_t lds = 0;
        const void* chosen = nullptr;
    };
    static thread_local LaunchChoice cache[4];
    static thread_local uint32_t cache_next = 0u;
    const LaunchChoice* hit = nullptr;
    for (const LaunchChoice& c : cache)
        if (c.fn == reinterpret_cast<const void*>(kernel_fn) && c.compact == reinterpret_cast<const void*>(compact_fn) && c.lds_scene == lds_scene &&
            c.item_bytes == item_bytes && c.keep_env == keep_env && c.pinned == pinned && c.t_max == t_max)
            hit = &c;
    if (hit != nullptr) {
        threads = hit->threads;
        per_cu = hit->per_cu;
        lds = hit->lds;
        g.pass_keep = hit->pass_keep;
        g.wave_bytes = hit->wave_bytes;
        chosen_fn = reinterpret_cast<void (*)(PathArgs, PersistArgs)>(const_cast<void*>(hit->chosen));
    } else {
        for (int variant = 0; variant < (compact_fn != nullptr ? 2 : 1); ++variant) {  // the compact per-wave area only if it keeps more waves on a CU
            const bool compact = variant == 1;
            void (*fn)(PathArgs, PersistArgs) = compact ? compact_fn : kernel_fn;
            // as many pass records as the LDS has room for without losing a wave: most first, so that a tie in waves keeps the most records
            for (uint32_t keep = accel ? keep_env : 0u;; keep = keep >= 8u ? keep - 8u : 0u) {
                // (large scenes: wave by wave -- a 6086-sphere scene's lists leave room for ten waves' areas, not for twelve: 640 threads, where
                // steps of 256 stopped at 512; the other kernels keep whole multiples of the four SIMDs)
                for (uint32_t t = 256u; t <= (compact ? 1024u : t_max); t += (accel && !shade_lds ? 64u : 256u)) {
                    if (pinned != 0u && t != pinned) continue;
                    const size_t need = lds_scene + kGroupLdsBytes + static_cast<size_t>(t / 64u) * wave_bytes_of(keep, compact);
                    if (need > kLdsPerCu) continue;
                    int blocks = 0;
                    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, static_cast<int>(t), need);
                    if (e != hipSuccess) return e;
                    if (blocks * static_cast<int>(t) > per_cu * static_cast<int>(threads)) {
                        threads = t;
                        per_cu = blocks;
                        lds = need;
                        g.pass_keep = keep;
                        g.wave_bytes = wave_bytes_of(keep, compact);
                        chosen_fn = fn;
                    }
                }
                if (keep == 0u) break;
            }
        }
        LaunchChoice& c = cache[cache_next++ % 4u];
        c.fn = reinterpret_cast<const void*>(kernel_fn); c.compact = reinterpret_cast<const void*>(compact_fn); c.lds_scene = lds_scene;
        c.item_bytes = item_bytes; c.keep_env = keep_env; c.pinned = pinned; c.t_max = t_max;
        c.threads = threads; c.per_cu = per_cu; c.lds = lds; c.pass_keep = g.pass_keep; c.wave_bytes = g.wave_bytes;
        c.chosen = reinterpret_cast<const void*>(chosen_fn);
    }
    if (debug_knob("RTIOW_DEBUG_LAUNCH"))  // (knobs builds: what the search chose -- tools/grid_stats.py)
        fprintf(stderr, "launch: %u spheres, %u-thread groups x %d per CU = %d waves per CU, %zu B of LDS a group (%zu the lists), %u pass records a wave%s\n", a.n, threads,
                per_cu, per_cu * static_cast<int>(threads / 64u), lds, lds_scene, g.pass_keep, chosen_fn == compact_fn && compact_fn != nullptr ? ", compact areas" : "");
    kernel_fn = chosen_fn;
    // camera paths a pass may make: one per lane, and no more than the records hold (those beyond pass_keep lie in the
    // work-list area: 42 of them, or all 64 in the two-level one)
    g.pass_min_idle = kPassMinLanes > g.pass_keep ? kPassMinLanes - g.pass_keep : 1u;
    // (the large-scene variant, the only one that looks at it: with records of its own it still waits for 16 idle slots -- a pass
    // of its two-level culls for a handful of rays costs more than the slots it fills; cover scenes of 1026 / 1938 / 3138 spheres
    // 5.78 -> 5.48, 6.73 -> 6.61, 7.52 -> 7.28 ms at 64 spp, tools/keep_ab.py)
    if (!shade_lds && g.pass_keep != 0u && g.pass_min_idle < 16u) g.pass_min_idle = 16u;
    if (const char* v = debug_knob("RTIOW_DEBUG_PASS_MIN_IDLE")) g.pass_min_idle = strtoul(v, nullptr, 10);  // tuning only
    g.pass_cap = g.pass_keep + item_bytes / kPassRecBytes;
    if (g.pass_cap > 64u) g.pass_cap = 64u;
    if (accel) {
        const bool cull = cone_setup(a, g);
        g.use_pass = kernel == KERNEL_CLUSTERED_PASS ||
                     a.spp >= (debug_knob("RTIOW_DEBUG_PASS_MIN_SPP") ? strtoul(debug_knob("RTIOW_DEBUG_PASS_MIN_SPP"), nullptr, 10) : kPassMinSpp) ? 1u : 0u;
        g.primary_all = (cull && !debug_knob("RTIOW_DEBUG_NO_CONE")) ? 0u : 1u;
    }
    if (threads == 0u) return hipErrorInvalidValue;  // rtSetScene's sphere limit keeps this from happening
    // persistent grid: fill the chip once; never more slots than samples
    unsigned long long grid = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256) * per_cu;
    const unsigned long long samples = static_cast<unsigned long long>(g.total_pix) * a.spp;
    const unsigned long long want_blocks = (samples + threads * kSlots - 1) / (threads * kSlots);
    if (grid > want_blocks) grid = want_blocks;
    if (const char* v = debug_knob("RTIOW_DEBUG_GRID")) grid = strtoul(v, nullptr, 10);  // tuning only
    if (grid < 1) grid = 1;
    g.total_waves = static_cast<uint32_t>(grid) * (threads / 64u);
    // A pool is never more than a share (an eighth or a quarter; rounds 2-3: a quarter) of what a wave gets in all: pools are of one size to the end of a queue (no look
    // at the head), and a cheap scene asks for large ones -- the three-sphere frame of BASELINE config 2, 29 pixels per
    // wave, was dealt in pools of 40 and took 1.93 ms instead of 0.8 (RTIOW_DEBUG_POOL_SHARE: tuning only).
    {
        // Where a wave's share of the frame is large -- 8192 samples and more: the cover frame has 23 400, half of it 11 700 -- pools of
        // twice the work, but no more than an eighth of the share: fewer fetches (each two dependent atomics and ~40 bytes of fabric
        // traffic: see kPoolWork).  Where it is small -- a quarter or an eighth of the cover frame, a 16-spp frame, 300 x 200 x 10 spp
        // with its 146 samples per wave -- pools stay as they were: doubled they took one eighth of the cover frame from 1.02 to 1.09 ms,
        // capped at an eighth the 300 x 200 frame from 0.28 to 0.36 (profiles/r04_pool_rule.txt).
#ifndef RTIOW_STALE_SMALL_POOLS
#define RTIOW_STALE_SMALL_POOLS 1  // (-DRTIOW_STALE_SMALL_POOLS=0: A/B only)
#endif
        // (... and only in an order made for THIS view: a large pool of a chunk the last view took for sky is one wave's time late in the frame)
        const bool large_share = static_cast<unsigned long long>(g.total_pix) * a.spp / g.total_waves >= 8192ull &&
                                 !(RTIOW_STALE_SMALL_POOLS && a.order_stale != 0u);
        if (large_share && !debug_knob("RTIOW_DEBUG_POOL_PIX")) {
            g.pool_pix = 2u * pool_samples / a.spp;
            g.pool_pix = g.pool_pix < 1u ? 1u : (g.pool_pix > 256u ? 256u : g.pool_pix);
        }
        uint32_t share = large_share ? 8u : 4u;
        if (const char* v = debug_knob("RTIOW_DEBUG_POOL_SHARE")) share = strtoul(v, nullptr, 10);
        const uint32_t cap = share ? g.total_pix / (g.total_waves * share) : ~0u;
        if (g.pool_pix > (cap < 1u ? 1u : cap)) g.pool_pix = cap < 1u ? 1u : cap;
    }
    // Whole-chunk pools (one store per 128-byte line of the frame) until a queue is down to about 24 pixels per wave that
    // draws from it: when the switch to small pools comes, every wave still holds half a chunk on average, and the small
    // pools have to fill the time until the last of them is through (RTIOW_DEBUG_CHUNK_UNTIL: tuning only).
    g.chunk_pool = g.pool_pix / kChunkPix * kChunkPix;
    if (g.chunk_pool < kChunkPix) g.chunk_pool = kChunkPix;
    g.chunk_until = (g.total_waves / 8u + 1u) * 24u + g.chunk_pool;
    // ... and only for frames with at least 256 pixels per wave: a chunk is a run of pixels of ONE wave's time, the dearest
    // of them (glass, ten times the average) ten chunks' worth; on one eighth of the cover frame (39 pixels per wave) whole
    // chunks made the tile take 3.9 ms instead of 1.7.  (256 pixels whatever the chunk: with the 16-pixel chunks of round 5 and the rule
    // left at "eight chunks" the cover frame -- 234 pixels per wave -- took whole chunks, and an orbiting camera, whose order is the last
    // view's, paid 17.7 % over its own views standing still instead of 3.6 %.)
    if (g.total_pix / 256u < g.total_waves) g.chunk_until = ~0u;
    // ... and only in the order of the last frame's costs, where the queue ends on the cheapest chunks.  In natural order
    // a dear chunk may come last and its wave finish alone: the first frame of a shape took 10.9-11.1 ms with whole
    // chunks against 10.5 without (9.6 once the order is there).
    if (a.chunk_order == nullptr) g.chunk_until = ~0u;
    if (const char* v = debug_knob("RTIOW_DEBUG_CHUNK_UNTIL")) g.chunk_until = strtoul(v, nullptr, 10);
    // ... and where no whole chunks are handed out -- small frames -- the head of an ORDERED queue, its dearest eighth, goes out
    // pixel by pixel (see the fetch): a queue has total_pix / 8 pixels (RTIOW_DEBUG_FINE_DIV: tuning only; 0 switches it off)
    {
        uint32_t div = 8u;
        if (const char* v = debug_knob("RTIOW_DEBUG_FINE_DIV")) div = strtoul(v, nullptr, 10);
        g.fine_pix = 128u / a.spp < 1u ? 1u : 128u / a.spp;  // (a 1-spp frame: 128 pixels -- more than its pools hold, i.e. no fine dealing)
        // (small frames only -- fewer than 48 pixels per wave, one eighth of the cover frame has 29: on larger ones no single pool
        // is the frame's time, and every fetch is two dependent atomics and ~40 bytes of fabric traffic)
        const bool small_frame = g.total_pix / g.total_waves < 48u || (RTIOW_STALE_SMALL_POOLS && a.order_stale != 0u);
        g.fine_until = (g.chunk_until == ~0u && a.chunk_order != nullptr && div != 0u && g.pool_pix > g.fine_pix && small_frame) ? g.total_pix / 8u / div : 0u;
    }
    hipLaunchKernelGGL(kernel_fn, dim3(static_cast<uint32_t>(grid)), dim3(threads), lds, stream, a, g);
    return hipGetLastError();
}

hipError_t launch_ch_sky_steps(float lo, float hi, RtChSkyStep* out, uint32_t cap, uint32_t* count, hipStream_t stream) {
    auto key_of = [](float x) {
        uint32_t b;
        std::memcpy(&b, &x, 4);
        return (b & 0x80000000u) ? -static_cast<long long>(b & 0x7FFFFFFFu) : static_cast<long long>(b);
    };
    const long long k0 = key_of(lo), k1 = key_of(hi);
    if (k1 < k0) return hipErrorInvalidValue;
    const unsigned long long n = static_cast<unsigned long long>(k1 - k0) + 1ull;
    hipLaunchKernelGGL(ch_sky_steps_kernel, dim3(static_cast<uint32_t>((n + 255ull) / 256ull)), dim3(256), 0, stream, k0, n, out, cap, count);
    return hipGetLastError();
}

hipError_t launch_unary_scan(uint32_t fn, float lo, float hi, unsigned long long* bad, uint32_t* first, uint32_t cap, hipStream_t stream) {
    uint32_t b0, b1;
    std::memcpy(&b0, &lo, 4);
    std::memcpy(&b1, &hi, 4);
    if ((b0 | b1) & 0x80000000u || b1 < b0) return hipErrorInvalidValue;  // positive floats, ascending
    const unsigned long long n = static_cast<unsigned long long>(b1 - b0) + 1ull;
    hipLaunchKernelGGL(unary_scan_kernel, dim3(static_cast<uint32_t>((n + 255ull) / 256ull)), dim3(256), 0, stream, fn, b0, n, bad, first, cap);
    return hipGetLastError();
}

hipError_t launch_arith(uint32_t op, const float* a, const float* b, const float* c, float* out,
                        uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(arith_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, op, a, b, c,
                       out, n);
    return hipGetLastError();
}

#endif  // RTIOW_TU_SMALL_CLUSTERED / RTIOW_TU_LARGE_CLUSTERED

}  // namespace rtiow
