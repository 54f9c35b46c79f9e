// rtiow_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the
// per-pixel path-tracing hot path.
//
//   ch_kernel            raytrace05.comp / raytrace06.comp (the reference's two
//                        compute shaders), one lane per pixel, packed RGBA8 store
//   path_pixel_kernel    PATH mode v1: one lane per pixel, spp + bounce loops inside,
//                        sphere list staged in LDS once per workgroup
//
// Arithmetic contract: binary32, round-to-nearest-even, denormals kept, only
// + - * / sqrt and fma; compiled with -ffp-contract=off so a fused multiply-add
// exists exactly where __builtin_fmaf is written (hipcc's default correctly
// rounded fp32 divide/sqrt stay on).  tests/ check every kernel bit-for-bit
// against the independent CPU restatement in oracle/.
#include <hip/hip_runtime.h>

#include "rtiow_device.h"
#include "rtiow_rng.h"

namespace rtiow {
namespace {

#define DI __device__ __forceinline__

struct f3 {
    float x, y, z;
};

DI f3 mk(float x, float y, float z) { return f3{x, y, z}; }
DI float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DI float dot3(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
DI f3 unit3(f3 a) {
    const float k = 1.0f / __builtin_sqrtf(dot3(a, a));
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

DI uint32_t resolve_pixel(f3 sum, uint32_t spp, uint32_t quantiser) {
    const float scale = 1.0f / static_cast<float>(spp);
    const float r = __builtin_sqrtf(scale * sum.x);
    const float g = __builtin_sqrtf(scale * sum.y);
    const float b = __builtin_sqrtf(scale * sum.z);
    if (quantiser == RT_QUANT_BOOK) return pack_rgb(quant_book(r), quant_book(g), quant_book(b));
    return pack_rgb(quant_unorm8(r), quant_unorm8(g), quant_unorm8(b));
}

DI uint32_t tile_global_row(uint32_t lr, uint32_t row_block, uint32_t rank, uint32_t count) {
    if (count <= 1) return lr;
    return ((lr / row_block) * count + rank) * row_block + lr % row_block;
}

// ============================================================================
// CH05 / CH06 — the reference's shaders
// ============================================================================

// GLSL dot/normalize without contraction (left-to-right)
DI float gdot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DI f3 gnormalize(f3 a) {
    const float l = __builtin_sqrtf(gdot(a, a));
    return mk(a.x / l, a.y / l, a.z / l);
}

// 16x16 workgroup as raytrace06.comp:2; ceil-div grid with a bounds check
// (fixes the fixed 64x64 / floor-div dispatches of main.cpp:321 and
// RTCHAP05 main.cpp:306).  One packed 32-bit store per lane: a wave writes
// four 64-byte row segments.
__global__ __launch_bounds__(256) void ch_kernel(ChArgs a) {
    const uint32_t tiles_x = (a.width + 15u) / 16u;
    const uint32_t gx = (blockIdx.x % tiles_x) * 16u + (threadIdx.x & 15u);
    const uint32_t gy = (blockIdx.x / tiles_x) * 16u + (threadIdx.x >> 4);
    if (gx >= a.width || gy >= a.height) return;

    // raytrace06.comp:53-61
    const f3 origin = mk(0.0f, 0.0f, 0.0f);
    const f3 horizontal = mk(a.ubo.viewportWidth, 0.0f, 0.0f);
    const f3 vertical = mk(0.0f, a.ubo.viewportHeight, 0.0f);
    f3 llc;
    llc.x = ((origin.x - horizontal.x / 2) - vertical.x / 2) - 0.0f;
    llc.y = ((origin.y - horizontal.y / 2) - vertical.y / 2) - 0.0f;
    llc.z = ((origin.z - horizontal.z / 2) - vertical.z / 2) - a.ubo.focalLength;
    const float u = static_cast<float>(gx) / (a.ubo.imageWidth - 1);
    const float v = static_cast<float>(gy) / (a.ubo.imageHeight - 1);
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

// ============================================================================
// PATH mode building blocks (BUILD-SPEC: SURVEY.md section 9)
// ============================================================================

struct Path {
    f3 o, du, att;  // origin, unit direction, attenuation product
    Pcg rng;
    DI Path() : rng(0u) {}
};

DI f3 random_in_unit_sphere(Pcg& rng) {
    for (;;) {
        f3 p;
        p.x = rng.symmetric();
        p.y = rng.symmetric();
        p.z = rng.symmetric();
        if (dot3(p, p) < 1.0f) return p;
    }
}

// one camera sample of pixel (i, j): SURVEY 9.4 / 9.5
DI void camera_path(const PathArgs& a, uint32_t i, uint32_t j, uint32_t sample, Path& p) {
    const RtCamera& c = a.cam;
    p.rng = Pcg(a.seed, j * a.width + i, sample);
    const float u = (static_cast<float>(i) + p.rng.uniform()) / static_cast<float>(a.width - 1);
    const float v = (static_cast<float>(j) + p.rng.uniform()) / static_cast<float>(a.height - 1);
    f3 off = mk(0.0f, 0.0f, 0.0f);
    if (c.lens_radius > 0.0f) {  // wave-uniform
        float dx, dy;
        for (;;) {  // random_in_unit_disk
            dx = p.rng.symmetric();
            dy = p.rng.symmetric();
            if (fma_(dy, dy, dx * dx) < 1.0f) break;
        }
        const float rdx = c.lens_radius * dx, rdy = c.lens_radius * dy;
        off = mk(fma_(c.v[0], rdy, c.u[0] * rdx), fma_(c.v[1], rdy, c.u[1] * rdx),
                 fma_(c.v[2], rdy, c.u[2] * rdx));
    }
    p.o = mk(c.origin[0] + off.x, c.origin[1] + off.y, c.origin[2] + off.z);
    f3 d;
    d.x = fma_(v, c.vertical[0], fma_(u, c.horizontal[0], c.lower_left[0])) - c.origin[0] - off.x;
    d.y = fma_(v, c.vertical[1], fma_(u, c.horizontal[1], c.lower_left[1])) - c.origin[1] - off.y;
    d.z = fma_(v, c.vertical[2], fma_(u, c.horizontal[2], c.lower_left[2])) - c.origin[2] - off.z;
    p.du = unit3(d);
    p.att = mk(1.0f, 1.0f, 1.0f);
}

DI f3 sky_radiance(const Path& p) {  // raytrace06.comp:45-47, direction already unit
    const float t = 0.5f * (p.du.y + 1.0f);
    const float k = 1.0f - t;
    const f3 c = mk(fma_(t, 0.5f, k), fma_(t, 0.7f, k), fma_(t, 1.0f, k));
    return mk(p.att.x * c.x, p.att.y * c.y, p.att.z * c.z);
}

// Hit at distance s on sphere `idx`: scatter per material.
// Returns false when the path is absorbed (radiance 0).
DI bool scatter(const PathArgs& a, int idx, float s, Path& p) {
    const float4 sp = a.spheres[idx];  // cx, cy, cz, radius
    const RtMaterial m = a.materials[idx];
    const f3 hit = mk(fma_(s, p.du.x, p.o.x), fma_(s, p.du.y, p.o.y), fma_(s, p.du.z, p.o.z));
    const f3 outward = mk((hit.x - sp.x) / sp.w, (hit.y - sp.y) / sp.w, (hit.z - sp.z) / sp.w);
    const float dn = dot3(p.du, outward);
    const bool front = dn < 0.0f;
    const f3 n = front ? outward : mk(-outward.x, -outward.y, -outward.z);
    f3 dir;
    if (m.kind == RT_MAT_LAMBERTIAN) {
        const f3 rv = unit3(random_in_unit_sphere(p.rng));
        dir = mk(n.x + rv.x, n.y + rv.y, n.z + rv.z);
        if (__builtin_fabsf(dir.x) < 1e-8f && __builtin_fabsf(dir.y) < 1e-8f &&
            __builtin_fabsf(dir.z) < 1e-8f)
            dir = n;
        p.att = mk(p.att.x * m.albedo[0], p.att.y * m.albedo[1], p.att.z * m.albedo[2]);
    } else if (m.kind == RT_MAT_METAL) {
        const float k2 = 2.0f * dot3(p.du, n);
        const f3 refl = mk(fma_(-k2, n.x, p.du.x), fma_(-k2, n.y, p.du.y), fma_(-k2, n.z, p.du.z));
        dir = refl;
        if (m.fuzz > 0.0f) {
            const f3 rs = random_in_unit_sphere(p.rng);
            dir = mk(fma_(m.fuzz, rs.x, refl.x), fma_(m.fuzz, rs.y, refl.y),
                     fma_(m.fuzz, rs.z, refl.z));
        }
        if (!(dot3(dir, n) > 0.0f)) return false;
        p.att = mk(p.att.x * m.albedo[0], p.att.y * m.albedo[1], p.att.z * m.albedo[2]);
    } else {
        const float ratio = front ? (1.0f / m.ior) : m.ior;
        const float nd = -dot3(p.du, n);
        const float cosv = (nd < 1.0f) ? nd : 1.0f;
        const float sinv = __builtin_sqrtf(fma_(-cosv, cosv, 1.0f));
        bool reflect = ratio * sinv > 1.0f;
        if (!reflect) {  // Schlick; (1-cos)^5 by repeated multiply, never powf
            float r0 = (1.0f - ratio) / (1.0f + ratio);
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
            const float par = -__builtin_sqrtf(__builtin_fabsf(1.0f - dot3(perp, perp)));
            dir = mk(fma_(par, n.x, perp.x), fma_(par, n.y, perp.y), fma_(par, n.z, perp.z));
        }
    }
    p.o = hit;
    p.du = unit3(dir);
    return true;
}

// Stage the sphere list into LDS as {cx, cy, cz, r*r}: 16 B per sphere
// (485 -> 7.6 KiB, 4096 -> 64 KiB; gfx950 has 160 KiB per CU).
DI void stage_spheres(const PathArgs& a, float4* lds) {
    for (uint32_t i = threadIdx.x; i < a.n; i += blockDim.x) {
        float4 s = a.spheres[i];
        s.w = s.w * s.w;
        lds[i] = s;
    }
    __syncthreads();
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
        const float sq = __builtin_sqrtf(disc);
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

// ============================================================================
// PATH v1: one lane per pixel
// ============================================================================
__global__ __launch_bounds__(256) void path_pixel_kernel(PathArgs a) {
    extern __shared__ float4 lds_spheres[];
    __shared__ unsigned int blk_paths, blk_segments;
    if (threadIdx.x == 0) {
        blk_paths = 0;
        blk_segments = 0;
    }
    stage_spheres(a, lds_spheres);

    const uint32_t tiles_x = (a.width + 15u) / 16u;
    const uint32_t i = (blockIdx.x % tiles_x) * 16u + (threadIdx.x & 15u);
    const uint32_t lr = (blockIdx.x / tiles_x) * 16u + (threadIdx.x >> 4);
    uint32_t n_paths = 0, n_segments = 0;
    if (i < a.width && lr < a.local_rows) {
        const uint32_t j = tile_global_row(lr, a.row_block, a.tile_rank, a.tile_count);
        f3 sum = mk(0.0f, 0.0f, 0.0f);
        for (uint32_t s0 = 0; s0 < a.spp; s0 += a.chunk_spp) {
            const uint32_t s1 = (s0 + a.chunk_spp < a.spp) ? s0 + a.chunk_spp : a.spp;
            f3 part = mk(0.0f, 0.0f, 0.0f);
            for (uint32_t s = s0; s < s1; ++s) {
                Path p;
                camera_path(a, i, j, s, p);
                ++n_paths;
                f3 rad = mk(0.0f, 0.0f, 0.0f);
                for (uint32_t depth = 0; depth < a.max_depth; ++depth) {
                    ++n_segments;
                    float dist;
                    const int idx = closest_hit_simple(lds_spheres, a.n, p, dist);
                    if (idx < 0) {
                        rad = sky_radiance(p);
                        break;
                    }
                    if (!scatter(a, idx, dist, p)) break;
                }
                part = mk(part.x + rad.x, part.y + rad.y, part.z + rad.z);
            }
            sum = mk(sum.x + part.x, sum.y + part.y, sum.z + part.z);
        }
        a.dst[static_cast<size_t>(lr) * a.dst_stride + i] = resolve_pixel(sum, a.spp, a.quantiser);
    }
    atomicAdd(&blk_paths, n_paths);
    atomicAdd(&blk_segments, n_segments);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&a.counters->paths, static_cast<unsigned long long>(blk_paths));
        atomicAdd(&a.counters->segments, static_cast<unsigned long long>(blk_segments));
    }
}

// ============================================================================
// PATH v2: persistent waves, chunk queue, ballot refill, candidate bitmasks
// ============================================================================
//
// Work item = one chunk: chunk_spp consecutive samples of one pixel, summed
// sequentially into one partial sum (the pixel is the sequential sum of its
// partial sums, resolved by resolve_kernel: the order is part of the spec, so
// the frame does not depend on which lane traced what).  Every lane owns R
// path slots.  One loop iteration traces exactly one segment per live slot:
//
//   refill   slots whose path ended start the next sample of their chunk; slots
//            whose chunk is finished store its partial sum and pull a new chunk:
//            __ballot of the needy lanes, ONE atomicAdd per wave on the queue
//            head, ids handed out by mbcnt prefix — dead lanes are refilled
//            immediately, so the sphere loop always runs with full waves.
//   trace    all lanes walk the LDS sphere list in lock-step (broadcast reads),
//            branch-free: 11 VALU per ray-sphere test, the sign bit of the
//            discriminant shifted into a per-lane candidate word by one
//            v_alignbit.  Only candidates (a handful per ray) take the
//            sqrt/root path, per lane, after each block of 32 spheres.
//   shade    miss -> sky into the partial sum; hit -> scatter by material.
//
// Order-independence: the closest hit is the minimum over spheres of each
// sphere's first root in (t_min, inf), ties to the lowest index — exactly what
// the oracle's sequential scan computes — so candidates may be examined in any
// grouping.

constexpr int kSlots = 2;          // path slots per lane
constexpr uint32_t kBlockSph = 32; // spheres per candidate word

struct Slot {
    Path p;
    f3 part;               // partial sum of the current chunk
    uint32_t pix;          // local pixel index
    uint32_t s, s_end;     // next sample, end of chunk
    uint32_t chunk;        // chunk id (index into partials)
    uint32_t depth;
    bool active, has_chunk;
};

DI void examine_candidate(const float4* lds, uint32_t j, uint32_t n, const Path& p, float& best,
                          int& best_i) {
    if (j >= n) return;  // list padding
    const float4 s = lds[j];
    const float ocx = p.o.x - s.x, ocy = p.o.y - s.y, ocz = p.o.z - s.z;
    const float hb = fma_(ocz, p.du.z, fma_(ocy, p.du.y, ocx * p.du.x));
    const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
    const float disc = fma_(hb, hb, -cc);
    if (__builtin_signbit(disc) || disc != disc) return;
    const float sq = __builtin_sqrtf(disc);
    float root = -hb - sq;
    if (!(root > kTMin && root < best)) {
        root = -hb + sq;
        if (!(root > kTMin && root < best)) return;
    }
    // ascending j within a slot: a tie keeps the earlier (lower) index
    best = root;
    best_i = static_cast<int>(j);
}

template <int R>
DI void trace_slots(const float4* lds, uint32_t n_pad, uint32_t n, Slot (&sl)[R], float (&best)[R],
                    int (&best_i)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        best[r] = __builtin_inff();
        best_i[r] = -1;
    }
    for (uint32_t base = 0; base < n_pad; base += kBlockSph) {
        uint32_t miss[R];
#pragma unroll
        for (int r = 0; r < R; ++r) miss[r] = 0u;
#pragma unroll 8
        for (uint32_t j = 0; j < kBlockSph; ++j) {
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
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t cand = sl[r].active ? ~miss[r] : 0u;
            while (cand) {
                const uint32_t bit = static_cast<uint32_t>(__builtin_clz(cand));
                cand &= ~(0x80000000u >> bit);
                examine_candidate(lds, base + bit, n, sl[r].p, best[r], best_i[r]);
            }
        }
    }
}

__global__ __launch_bounds__(1024) void path_persistent_kernel(PathArgs a, uint32_t n_pad,
                                                              uint32_t chunks_per_pixel,
                                                              unsigned long long total_chunks) {
    extern __shared__ float4 lds_spheres[];
    for (uint32_t i = threadIdx.x; i < n_pad; i += blockDim.x) {
        float4 s;
        if (i < a.n) {
            s = a.spheres[i];
            s.w = s.w * s.w;
        } else {
            s = make_float4(0.0f, 0.0f, 0.0f, -1.0f);  // padding: disc = hb^2 - |o|^2 - 1 < 0
        }
        lds_spheres[i] = s;
    }
    __syncthreads();

    Slot sl[kSlots];
#pragma unroll
    for (int r = 0; r < kSlots; ++r) {
        sl[r].active = false;
        sl[r].has_chunk = false;
        sl[r].part = mk(0.0f, 0.0f, 0.0f);
        sl[r].pix = sl[r].s = sl[r].s_end = sl[r].chunk = sl[r].depth = 0u;
        sl[r].p.o = sl[r].p.du = sl[r].p.att = mk(0.0f, 0.0f, 0.0f);
    }
    bool exhausted = false;  // wave-uniform: the queue has been drained
    uint32_t n_paths = 0, n_segments = 0;

    for (;;) {
        // ---- refill ---------------------------------------------------------
#pragma unroll
        for (int r = 0; r < kSlots; ++r) {
            Slot& q = sl[r];
            bool need = !q.active && (!q.has_chunk || q.s == q.s_end);
            if (need && q.has_chunk) {
                a.partials[q.chunk] = make_float4(q.part.x, q.part.y, q.part.z, 0.0f);
                q.has_chunk = false;
            }
            if (!exhausted) {
                const unsigned long long mask = __ballot(need);
                if (mask != 0ull) {
                    const uint32_t want = static_cast<uint32_t>(__popcll(mask));
                    unsigned long long base = 0ull;
                    if (__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u)) == 0u && need)
                        base = atomicAdd(&a.counters->queue_head, static_cast<unsigned long long>(want));
                    // broadcast from the first needy lane
                    const int leader = __builtin_ctzll(mask);
                    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(base), leader);
                    const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(base >> 32), leader);
                    base = (static_cast<unsigned long long>(hi) << 32) | lo;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
                        static_cast<uint32_t>(mask >> 32),
                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
                    const unsigned long long id = base + rank;
                    if (need && id < total_chunks) {
                        q.chunk = static_cast<uint32_t>(id);
                        q.pix = static_cast<uint32_t>(id / chunks_per_pixel);
                        const uint32_t k = static_cast<uint32_t>(id % chunks_per_pixel);
                        q.s = k * a.chunk_spp;
                        q.s_end = (q.s + a.chunk_spp < a.spp) ? q.s + a.chunk_spp : a.spp;
                        q.part = mk(0.0f, 0.0f, 0.0f);
                        q.has_chunk = true;
                    }
                    if (base + want >= total_chunks) exhausted = true;
                }
            }
            if (!q.active && q.has_chunk) {  // next sample of the chunk
                const uint32_t lr = q.pix / a.width, i = q.pix % a.width;
                const uint32_t j = tile_global_row(lr, a.row_block, a.tile_rank, a.tile_count);
                camera_path(a, i, j, q.s, q.p);
                ++q.s;
                q.depth = 0u;
                q.active = true;
                ++n_paths;
            }
        }
        bool any_active = false;
#pragma unroll
        for (int r = 0; r < kSlots; ++r) any_active = any_active || sl[r].active;
        if (__ballot(any_active) == 0ull) break;  // queue drained and every path finished

        // ---- trace ----------------------------------------------------------
        float best[kSlots];
        int best_i[kSlots];
        trace_slots<kSlots>(lds_spheres, n_pad, a.n, sl, best, best_i);

        // ---- shade ----------------------------------------------------------
#pragma unroll
        for (int r = 0; r < kSlots; ++r) {
            Slot& q = sl[r];
            if (!q.active) continue;
            ++n_segments;
            if (best_i[r] < 0) {
                const f3 rad = sky_radiance(q.p);
                q.part = mk(q.part.x + rad.x, q.part.y + rad.y, q.part.z + rad.z);
                q.active = false;
            } else if (!scatter(a, best_i[r], best[r], q.p)) {
                q.active = false;  // absorbed: radiance 0
            } else if (++q.depth >= a.max_depth) {
                q.active = false;  // depth exhausted: radiance 0
            }
        }
    }

    // one counter update per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_paths += __shfl_down(n_paths, off);
        n_segments += __shfl_down(n_segments, off);
    }
    if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(&a.counters->paths, static_cast<unsigned long long>(n_paths));
        atomicAdd(&a.counters->segments, static_cast<unsigned long long>(n_segments));
    }
}

// partial sums -> RGBA8, one lane per pixel (coalesced 4-byte stores: 256 B per wave)
__global__ __launch_bounds__(256) void resolve_kernel(PathArgs a, uint32_t chunks_per_pixel) {
    const uint32_t lp = blockIdx.x * blockDim.x + threadIdx.x;
    if (lp >= a.local_rows * a.width) return;
    f3 sum = mk(0.0f, 0.0f, 0.0f);
    const float4* part = a.partials + static_cast<size_t>(lp) * chunks_per_pixel;
    for (uint32_t k = 0; k < chunks_per_pixel; ++k) {
        const float4 v = part[k];
        sum = mk(sum.x + v.x, sum.y + v.y, sum.z + v.z);
    }
    const uint32_t lr = lp / a.width, i = lp % a.width;
    a.dst[static_cast<size_t>(lr) * a.dst_stride + i] = resolve_pixel(sum, a.spp, a.quantiser);
}

// ============================================================================
// arithmetic conformance probe (tests/test_gpu_arith.py)
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
        default: break;
    }
    out[i] = r;
}

}  // namespace

hipError_t launch_ch(const ChArgs& a, hipStream_t stream) {
    const uint32_t tiles = ((a.width + 15u) / 16u) * ((a.height + 15u) / 16u);
    hipLaunchKernelGGL(ch_kernel, dim3(tiles), dim3(256), 0, stream, a);
    return hipGetLastError();
}

static uint32_t chunks_per_pixel(const PathArgs& a) { return (a.spp + a.chunk_spp - 1u) / a.chunk_spp; }

static bool use_persistent(uint32_t kernel) { return kernel != KERNEL_PIXEL; }

size_t path_partials_bytes(const PathArgs& a, uint32_t kernel) {
    if (!use_persistent(kernel)) return 0;
    return static_cast<size_t>(a.local_rows) * a.width * chunks_per_pixel(a) * sizeof(float4);
}

hipError_t launch_path(const PathArgs& a, uint32_t kernel, int num_cus, hipStream_t stream) {
    if (!use_persistent(kernel)) {
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
    const uint32_t n_pad = (a.n + kBlockSph - 1u) / kBlockSph * kBlockSph;
    const size_t lds = static_cast<size_t>(n_pad) * sizeof(float4);
    // one LDS copy of the list per workgroup: small lists -> 256-thread groups (8 per CU);
    // large lists -> 1024-thread groups so 16 waves share one copy
    const uint32_t threads = lds <= 16u * 1024u ? 256u : 1024u;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(path_persistent_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, path_persistent_kernel, static_cast<int>(threads), lds);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    const uint32_t cpp = chunks_per_pixel(a);
    const unsigned long long total = static_cast<unsigned long long>(a.local_rows) * a.width * cpp;
    // persistent grid: fill the chip once; never more lanes than work items
    unsigned long long want_blocks = (total + threads * kSlots - 1) / (threads * kSlots);
    unsigned long long grid = static_cast<unsigned long long>(num_cus > 0 ? num_cus : 256) * per_cu;
    if (grid > want_blocks) grid = want_blocks;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(path_persistent_kernel, dim3(static_cast<uint32_t>(grid)), dim3(threads), lds, stream, a,
                       n_pad, cpp, total);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t px = a.local_rows * a.width;
    hipLaunchKernelGGL(resolve_kernel, dim3((px + 255u) / 256u), dim3(256), 0, stream, a, cpp);
    return hipGetLastError();
}

hipError_t launch_arith(uint32_t op, const float* a, const float* b, const float* c, float* out,
                        uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(arith_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, op, a, b, c,
                       out, n);
    return hipGetLastError();
}

}  // namespace rtiow
