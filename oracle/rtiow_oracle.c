/*
 * rtiow_oracle.c — CPU restatement of the per-pixel path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (vulkan-rtiow_amd/, the
 * C-ABI library, the C++ harness) may include, link or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as
 * the checker / the timed CPU baseline.
 *
 * What is pinned by the reference and what is not
 * -----------------------------------------------
 *  * RT_MODE_CH05 / RT_MODE_CH06 restate the reference's GLSL, statement by
 *    statement (RTCHAP05/RTCHAP05/Shaders/raytrace05.comp:21-61 and
 *    RTCHAP06/Shaders/raytrace06.comp:21-66), with the UBO of
 *    RTCHAP06/main.cpp:101-120 and the display flip of RTCHAP06/Shaders/rt.frag:8.
 *    They are pinned against the three result fixtures the reference holds --
 *    RTCHAP05/RTCHAP05/21986.jpg (silhouette bbox, hit count 167 084, corner
 *    colours), RTCHAP05/RTCHAP05/1728.jpg (the stretched 1024^2 image: ellipse
 *    bbox) and RT01/RT01/4068.jpg (sky) -- through tests/golden/ and the
 *    known-answer table of SURVEY.md section 8(c).  The GLSL cannot be compiled here (no glslc, Vulkan
 *    loader or ICD in the image), so "pinned by source text + that fixture".
 *  * RT_MODE_PATH (hittable list, lambertian/metal/dielectric, multi-sample
 *    accumulate, cover scene) DOES NOT EXIST in the reference (SURVEY.md
 *    section 0.1).  It follows the public RTIOW book as recorded in SURVEY.md
 *    section 9; this file is its authority.  PARITY UNPINNED for this mode.
 *    (What can be checked without the reference is checked analytically, by
 *    computations that share nothing with this file and hold for it and for
 *    every GPU kernel: tests/furnace.py (single bodies, head-on rays, a hall of
 *    non-absorbing bodies with hollow glass), tests/mirrors.py (reflection chains
 *    among several spheres), tests/glass.py (refraction directions and Schlick
 *    probabilities), tests/lambert.py (the diffuse distribution),
 *    tests/fuzzmetal.py (fuzz and absorption), tests/defocus.py (the thin lens).)
 *
 * Arithmetic contract (shared with the HIP kernels, which are written
 * independently of this file): IEEE-754 binary32, round-to-nearest-even,
 * denormals kept, only + - * / sqrt and fma, every operation in the order
 * written here, fused multiply-add exactly where fmaf() appears and nowhere
 * else (build with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rtiow.h"

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(v3 a, float k) { return V(a.x * k, a.y * k, a.z * k); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* dot := fma(az,bz, fma(ay,by, ax*bx)) */
static inline float vdot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
/* unit(a) := a * (1/sqrt(dot(a,a))) */
static inline v3 vunit(v3 a) {
    float k = 1.0f / sqrtf(vdot(a, a));
    return vscale(a, k);
}

/* ---- RNG: PCG-RXS-M-XS-32 stream per (seed, pixel, sample) ---------------
 * BUILD-SPEC (hard part 2): the book's sequential
 * rand() cannot be reproduced on a GPU. */
static inline uint32_t pcg_step(uint32_t s) { return s * 747796405u + 2891336453u; }
static inline uint32_t pcg_out(uint32_t s) {
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
static inline uint32_t rng_init(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t s = pcg_out(pcg_step(seed));
    s = pcg_out(pcg_step(s ^ pixel));
    s = pcg_out(pcg_step(s ^ sample));
    return s;
}
static inline float randf(uint32_t* st) { /* [0,1), 24 bits */
    *st = pcg_step(*st);
    return (float)(pcg_out(*st) >> 8) * 0x1p-24f;
}

/* Rejection-free sampling (BUILD-SPEC).  The book draws points by rejection from a
 * cube/square; on a 64-wide SIMD the trip count of a rejection loop is the maximum
 * over the lanes, so the spec uses closed forms built from + - * sqrt and fma only.
 * sincos_2pi: quadrant from the top two bits of u, then two polynomials in
 * f = frac(4u) for sin(pi/2 f), cos(pi/2 f) (max abs error 2e-7), fma-only so that
 * CPU and GPU agree bit for bit. */
static const float kS0 = 0x1.921fb6p+0f, kS1 = -0x1.4abbc4p-1f, kS2 = 0x1.4668f0p-4f,
                   kS3 = -0x1.32533cp-8f, kS4 = 0x1.3e15f6p-13f;
static const float kC0 = 0x1.fffffep-1f, kC1 = -0x1.3bd3a6p+0f, kC2 = 0x1.03bd02p-2f,
                   kC3 = -0x1.54f5dcp-6f, kC4 = 0x1.c1ecap-11f;

static void sincos_2pi(float u, float* c, float* s) { /* u in [0,1) */
    float t = 4.0f * u;          /* exact */
    int q = (int)t;              /* 0..3 */
    float f = t - (float)q;      /* exact */
    float f2 = f * f;
    float sp = f * fmaf(f2, fmaf(f2, fmaf(f2, fmaf(f2, kS4, kS3), kS2), kS1), kS0);
    float cp = fmaf(f2, fmaf(f2, fmaf(f2, fmaf(f2, kC4, kC3), kC2), kC1), kC0);
    switch (q) {
        case 0: *c = cp; *s = sp; break;
        case 1: *c = -sp; *s = cp; break;
        case 2: *c = -cp; *s = -sp; break;
        default: *c = sp; *s = -cp; break;
    }
}

/* uniform direction: z uniform in (-1,1], azimuth uniform; two draws, no rejection */
static v3 random_unit_vector(uint32_t* st) {
    float u1 = randf(st);
    float u2 = randf(st);
    float z = 1.0f - 2.0f * u1;
    float r = sqrtf(fmaf(-z, z, 1.0f));
    float c, s;
    sincos_2pi(u2, &c, &s);
    return V(r * c, r * s, z);
}

/* uniform point in the unit disk: radius sqrt(u1), azimuth 2 pi u2 */
static void random_in_unit_disk(uint32_t* st, float* dx, float* dy) {
    float u1 = randf(st);
    float u2 = randf(st);
    float r = sqrtf(u1);
    float c, s;
    sincos_2pi(u2, &c, &s);
    *dx = r * c;
    *dy = r * s;
}

/* ---- quantisers (a5 / a10) ---------------------------------------------- */
static inline uint32_t quant_unorm8(float x) { /* rgba8 imageStore, raytrace06.comp:3,66 */
    float c = (x > 0.0f) ? (x < 1.0f ? x : 1.0f) : 0.0f; /* NaN -> 0 */
    return (uint32_t)(int)(c * 255.0f + 0.5f);
}
static inline uint32_t quant_book(float x) { /* book write_color: (int)(256*clamp(x,0,0.999)) */
    float c = (x > 0.0f) ? (x < 0.999f ? x : 0.999f) : 0.0f;
    return (uint32_t)(int)(256.0f * c);
}
static inline uint32_t pack_rgba(uint32_t r, uint32_t g, uint32_t b) {
    return r | (g << 8) | (b << 16); /* alpha byte 0: vec4(color,0.0), raytrace06.comp:66 */
}

/* ---- row tiling --------------------------------------------- */
uint32_t oracle_tile_row_count(uint32_t height, uint32_t row_block, uint32_t rank, uint32_t count) {
    if (count <= 1) return height;
    if (row_block == 0) row_block = 1;
    uint32_t n = 0;
    for (uint32_t r = 0; r < height; r++)
        if ((r / row_block) % count == rank) n++;
    return n;
}
uint32_t oracle_tile_global_row(uint32_t local_row, uint32_t row_block, uint32_t rank,
                                uint32_t count) {
    if (count <= 1) return local_row;
    if (row_block == 0) row_block = 1;
    return ((local_row / row_block) * count + rank) * row_block + local_row % row_block;
}

/* ======================================================================== */
/* CH05 / CH06: the reference shaders                                        */
/* ======================================================================== */

/* GLSL dot(): left-to-right sum of products, no contraction */
static inline float gdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* GLSL normalize(): v / sqrt(dot(v,v)) */
static inline v3 gnormalize(v3 a) {
    float l = sqrtf(gdot(a, a));
    return V(a.x / l, a.y / l, a.z / l);
}

static uint32_t ch_pixel(const RtUbo5* ubo, uint32_t mode, uint32_t gx, uint32_t gy) {
    /* raytrace06.comp:53-61 == raytrace05.comp:47-55 */
    v3 origin = V(0, 0, 0);
    v3 horizontal = V(ubo->viewportWidth, 0, 0);
    v3 vertical = V(0, ubo->viewportHeight, 0);
    v3 llc = vsub(vsub(vsub(origin, V(horizontal.x / 2, horizontal.y / 2, horizontal.z / 2)),
                       V(vertical.x / 2, vertical.y / 2, vertical.z / 2)),
                  V(0, 0, ubo->focalLength));
    float u = (float)gx / (ubo->imageWidth - 1);
    float v = (float)gy / (ubo->imageHeight - 1);
    v3 dir = vsub(vadd(vadd(llc, vscale(horizontal, u)), vscale(vertical, v)), origin);

    /* hitSphere, raytrace06.comp:21-33 / raytrace05.comp:21-30 */
    v3 center = V(0, 0, -1);
    float radius = 0.5f;
    v3 oc = vsub(origin, center);
    float a = gdot(dir, dir);
    float b = 2.0f * gdot(oc, dir);
    float c = gdot(oc, oc) - radius * radius;
    float disc = b * b - 4 * a * c;

    v3 col;
    int shaded = 0;
    if (mode == RT_MODE_CH05) {
        if (disc > 0) { col = V(1, 0, 0); shaded = 1; } /* raytrace05.comp:29,35-37 */
    } else {
        float t = (disc < 0) ? -1.0f : (-b - sqrtf(disc)) / (2.0f * a); /* :29-32 */
        if (t > 0.0f) {                                                    /* :39-43 */
            v3 r = vadd(origin, vscale(dir, t));
            v3 N = gnormalize(vsub(r, V(0.0f, 0.0f, -1)));
            col = V(0.5f * (N.x + 1), 0.5f * (N.y + 1), 0.5f * (N.z + 1));
            shaded = 1;
        }
    }
    if (!shaded) { /* raytrace06.comp:45-47 */
        v3 unit = gnormalize(dir);
        float t = 0.5f * (unit.y + 1.0f);
        col = vadd(vscale(V(1.0f, 1.0f, 1.0f), 1.0f - t), vscale(V(0.5f, 0.7f, 1.0f), t));
    }
    return pack_rgba(quant_unorm8(col.x), quant_unorm8(col.y), quant_unorm8(col.z));
}

int oracle_render_ubo(const RtUbo5* ubo, uint32_t mode, uint8_t* dst, size_t pitch) {
    if (!ubo || !dst || (mode != RT_MODE_CH05 && mode != RT_MODE_CH06)) return RT_ERR_INVALID;
    uint32_t W = (uint32_t)ubo->imageWidth, H = (uint32_t)ubo->imageHeight; /* main.cpp:106 */
    if (pitch < (size_t)W * 4) return RT_ERR_INVALID;
    for (uint32_t gy = 0; gy < H; gy++) {
        uint32_t* row = (uint32_t*)(dst + (size_t)gy * pitch);
        for (uint32_t gx = 0; gx < W; gx++) row[gx] = ch_pixel(ubo, mode, gx, gy);
    }
    return RT_OK;
}

/* ======================================================================== */
/* PATH mode (BUILD-SPEC)                                  */
/* ======================================================================== */

typedef struct {
    const RtSphere* sph;
    const RtMaterial* mat;
    uint32_t n;
    RtCamera cam;
    RtParams p;
} Scene;

#define T_MIN 0.001f

/* closest hit over the list; every ray direction du is unit length.
 * Per sphere i (SURVEY 9.2 in half-b form with a == 1):
 *   oc = o - c;  hb = dot(oc,du);  cc = fma(ocz,ocz,fma(ocy,ocy,fma(ocx,ocx,-r*r)))
 *   disc = fma(hb,hb,-cc);  candidate iff disc is not NaN and its sign bit is clear
 *   s = -hb - sqrt(disc); if !(T_MIN < s < best) s = -hb + sqrt(disc); same test
 * Closest s wins; ties keep the lowest index. */
static int hit_list(const Scene* sc, v3 o, v3 du, float* out_s) {
    float best = INFINITY;
    int best_i = -1;
    for (uint32_t i = 0; i < sc->n; i++) {
        const RtSphere* s = &sc->sph[i];
        float r2 = s->radius * s->radius;
        float ocx = o.x - s->cx, ocy = o.y - s->cy, ocz = o.z - s->cz;
        float hb = fmaf(ocz, du.z, fmaf(ocy, du.y, ocx * du.x));
        float cc = fmaf(ocz, ocz, fmaf(ocy, ocy, fmaf(ocx, ocx, -r2)));
        float disc = fmaf(hb, hb, -cc);
        if (signbit(disc) || disc != disc) continue;
        float sq = sqrtf(disc);
        float root = -hb - sq;
        if (!(root > T_MIN && root < best)) {
            root = -hb + sq;
            if (!(root > T_MIN && root < best)) continue;
        }
        best = root;
        best_i = (int)i;
    }
    *out_s = best;
    return best_i;
}

static inline v3 sky(v3 du) { /* raytrace06.comp:45-47 with du already unit */
    float t = 0.5f * (du.y + 1.0f);
    float k = 1.0f - t;
    return V(fmaf(t, 0.5f, k), fmaf(t, 0.7f, k), fmaf(t, 1.0f, k));
}

/* iterative ray_color; returns radiance, counts segments */
static v3 ray_color(const Scene* sc, v3 o, v3 du, uint32_t* st, uint64_t* segs) {
    v3 att = V(1.0f, 1.0f, 1.0f);
    for (uint32_t depth = 0; depth < sc->p.max_depth; depth++) {
        float s;
        (*segs)++;
        int i = hit_list(sc, o, du, &s);
        if (i < 0) {
            v3 c = sky(du);
            return V(att.x * c.x, att.y * c.y, att.z * c.z);
        }
        const RtSphere* sp = &sc->sph[i];
        const RtMaterial* m = &sc->mat[i];
        v3 p = V(fmaf(s, du.x, o.x), fmaf(s, du.y, o.y), fmaf(s, du.z, o.z));
        float inv_r = 1.0f / sp->radius; /* negative radius flips the normal (hollow glass) */
        v3 on = V((p.x - sp->cx) * inv_r, (p.y - sp->cy) * inv_r, (p.z - sp->cz) * inv_r);
        float dn = vdot(du, on);
        int front = dn < 0.0f;
        v3 n = front ? on : vneg(on);
        v3 dir;
        if (m->kind == RT_MAT_LAMBERTIAN) {
            v3 rv = random_unit_vector(st);
            dir = vadd(n, rv);
            if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = n;
            att = V(att.x * m->albedo[0], att.y * m->albedo[1], att.z * m->albedo[2]);
        } else if (m->kind == RT_MAT_METAL) {
            float k2 = 2.0f * vdot(du, n);
            v3 refl = V(fmaf(-k2, n.x, du.x), fmaf(-k2, n.y, du.y), fmaf(-k2, n.z, du.z));
            dir = refl;
            if (m->fuzz > 0.0f) { /* book v4: reflected + fuzz * random_unit_vector() */
                v3 rs = random_unit_vector(st);
                dir = V(fmaf(m->fuzz, rs.x, refl.x), fmaf(m->fuzz, rs.y, refl.y),
                        fmaf(m->fuzz, rs.z, refl.z));
            }
            if (!(vdot(dir, n) > 0.0f)) return V(0, 0, 0); /* absorbed */
            att = V(att.x * m->albedo[0], att.y * m->albedo[1], att.z * m->albedo[2]);
        } else { /* dielectric; attenuation 1 */
            float ratio = front ? (1.0f / m->ior) : m->ior;
            float nd = -vdot(du, n);
            float cosv = (nd < 1.0f) ? nd : 1.0f;
            float sinv = sqrtf(fmaf(-cosv, cosv, 1.0f));
            int reflect = ratio * sinv > 1.0f;
            if (!reflect) { /* Schlick */
                float r0 = (1.0f - ratio) / (1.0f + ratio);
                r0 = r0 * r0;
                float x = 1.0f - cosv;
                float x2 = x * x;
                float x5 = (x2 * x2) * x;
                float prob = fmaf(1.0f - r0, x5, r0);
                reflect = prob > randf(st);
            }
            if (reflect) {
                float k2 = 2.0f * vdot(du, n);
                dir = V(fmaf(-k2, n.x, du.x), fmaf(-k2, n.y, du.y), fmaf(-k2, n.z, du.z));
            } else {
                v3 perp = V(ratio * fmaf(cosv, n.x, du.x), ratio * fmaf(cosv, n.y, du.y),
                            ratio * fmaf(cosv, n.z, du.z));
                float par = -sqrtf(fabsf(1.0f - vdot(perp, perp)));
                dir = V(fmaf(par, n.x, perp.x), fmaf(par, n.y, perp.y), fmaf(par, n.z, perp.z));
            }
        }
        o = p;
        du = vunit(dir);
    }
    return V(0, 0, 0); /* depth exhausted */
}

/* one camera sample of pixel (i,j) */
static v3 sample_pixel(const Scene* sc, uint32_t i, uint32_t j, uint32_t s, uint64_t* segs) {
    const RtCamera* c = &sc->cam;
    uint32_t st = rng_init(sc->p.seed, j * sc->p.width + i, s);
    /* same /(W-1) convention as raytrace06.comp:57-58, as a multiply by the rounded reciprocal */
    float inv_wm1 = 1.0f / (float)(sc->p.width - 1), inv_hm1 = 1.0f / (float)(sc->p.height - 1);
    float u = ((float)i + randf(&st)) * inv_wm1;
    float v = ((float)j + randf(&st)) * inv_hm1;
    v3 off = V(0, 0, 0);
    if (c->lens_radius > 0.0f) {
        float dx, dy;
        random_in_unit_disk(&st, &dx, &dy);
        float rdx = c->lens_radius * dx, rdy = c->lens_radius * dy;
        off = V(fmaf(c->v[0], rdy, c->u[0] * rdx), fmaf(c->v[1], rdy, c->u[1] * rdx),
                fmaf(c->v[2], rdy, c->u[2] * rdx));
    }
    v3 o = V(c->origin[0] + off.x, c->origin[1] + off.y, c->origin[2] + off.z);
    v3 d;
    d.x = fmaf(v, c->vertical[0], fmaf(u, c->horizontal[0], c->lower_left[0])) - c->origin[0] - off.x;
    d.y = fmaf(v, c->vertical[1], fmaf(u, c->horizontal[1], c->lower_left[1])) - c->origin[1] - off.y;
    d.z = fmaf(v, c->vertical[2], fmaf(u, c->horizontal[2], c->lower_left[2])) - c->origin[2] - off.z;
    return ray_color(sc, o, vunit(d), &st, segs);
}

/* Multi-sample accumulate (a10).  BUILD-SPEC: every sample's radiance component is
 * converted to unsigned 32.32 fixed point (clamp to [0, 32768], multiply by 2^32 —
 * exact — and truncate) and the pixel is the INTEGER sum of its samples, so the sum
 * does not depend on the order or grouping in which samples are added (a GPU may
 * split a pixel's samples over lanes and combine them with integer atomics).
 * write_color: c = sqrt((float)sum * (1 / (spp * 2^32))), then the quantiser. */
static inline uint64_t to_fixed(float x) {
    float c = (x > 0.0f) ? (x < 32768.0f ? x : 32768.0f) : 0.0f; /* NaN -> 0 */
    return (uint64_t)(c * 4294967296.0f);
}

static uint32_t path_pixel(const Scene* sc, uint32_t i, uint32_t j, uint64_t* segs) {
    uint32_t spp = sc->p.spp;
    uint64_t sr = 0, sg = 0, sb = 0;
    for (uint32_t s = 0; s < spp; s++) {
        v3 c = sample_pixel(sc, i, j, s, segs);
        sr += to_fixed(c.x);
        sg += to_fixed(c.y);
        sb += to_fixed(c.z);
    }
    float scale = 1.0f / ((float)spp * 4294967296.0f);
    float r = sqrtf(scale * (float)sr), g = sqrtf(scale * (float)sg), b = sqrtf(scale * (float)sb);
    if (sc->p.quantiser == RT_QUANT_BOOK) return pack_rgba(quant_book(r), quant_book(g), quant_book(b));
    return pack_rgba(quant_unorm8(r), quant_unorm8(g), quant_unorm8(b));
}

/* Renders the rows of this tile (all rows when tile_count <= 1) into dst,
 * packed in ascending row order.  nthreads <= 0: all cores.  Returns RT_OK. */
int oracle_render(const RtSphere* spheres, const RtMaterial* materials, uint32_t n,
                  const RtCamera* cam, const RtParams* params, uint8_t* dst, size_t pitch,
                  int nthreads, uint64_t* out_segments) {
    if (!cam || !params || !dst) return RT_ERR_INVALID;
    if (params->mode != RT_MODE_PATH) return RT_ERR_INVALID;
    if (!spheres || !materials || n == 0) return RT_ERR_STATE;
    if (params->width < 2 || params->height < 2 || params->spp == 0 || params->spp > 65536) return RT_ERR_INVALID;
    if (pitch < (size_t)params->width * 4) return RT_ERR_INVALID;
    Scene sc;
    sc.sph = spheres; sc.mat = materials; sc.n = n; sc.cam = *cam; sc.p = *params;
    uint32_t rows = oracle_tile_row_count(params->height, params->row_block, params->tile_rank,
                                          params->tile_count);
    uint64_t total = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_num_procs();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : total)
#endif
    for (uint32_t lr = 0; lr < rows; lr++) {
        uint32_t j = oracle_tile_global_row(lr, params->row_block, params->tile_rank,
                                            params->tile_count);
        uint32_t* row = (uint32_t*)(dst + (size_t)lr * pitch);
        uint64_t segs = 0;
        for (uint32_t i = 0; i < params->width; i++) row[i] = path_pixel(&sc, i, j, &segs);
        total += segs;
    }
    if (out_segments) *out_segments = total;
    return RT_OK;
}

int oracle_num_procs(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* ======================================================================== */
/* host-side helpers restated (camera, scenes, writer)                       */
/* ======================================================================== */

int oracle_ubo_from_image(uint32_t width, uint32_t height, RtUbo5* out) {
    /* RTCHAP06/main.cpp:103-120 */
    float aspectRatio = (float)width / (float)height;
    out->imageWidth = (float)width;
    out->imageHeight = (float)width / aspectRatio;
    out->viewportHeight = 2.0f / aspectRatio;
    out->viewportWidth = 2.0f;
    out->focalLength = 1.0f;
    return RT_OK;
}

int oracle_camera_from_ubo(const RtUbo5* ubo, RtCamera* c) {
    memset(c, 0, sizeof *c);
    c->horizontal[0] = ubo->viewportWidth;
    c->vertical[1] = ubo->viewportHeight;
    c->lower_left[0] = 0.0f - ubo->viewportWidth / 2;
    c->lower_left[1] = 0.0f - ubo->viewportHeight / 2;
    c->lower_left[2] = 0.0f - ubo->focalLength;
    c->u[0] = 1.0f; c->v[1] = 1.0f; c->w[2] = 1.0f;
    c->lens_radius = 0.0f;
    return RT_OK;
}

int oracle_make_camera(const float from[3], const float at[3], const float vup[3], float vfov_deg,
                       float aspect, float aperture, float focus, RtCamera* c) {
    /* SURVEY 9.4; double on the host, rounded to float once at the end */
    double theta = (double)vfov_deg * 3.14159265358979323846 / 180.0;
    double h = tan(theta / 2.0);
    double vh = 2.0 * h, vw = (double)aspect * vh;
    double w[3], u[3], v[3];
    for (int k = 0; k < 3; k++) w[k] = (double)from[k] - (double)at[k];
    double wl = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int k = 0; k < 3; k++) w[k] /= wl;
    u[0] = (double)vup[1] * w[2] - (double)vup[2] * w[1];
    u[1] = (double)vup[2] * w[0] - (double)vup[0] * w[2];
    u[2] = (double)vup[0] * w[1] - (double)vup[1] * w[0];
    double ul = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int k = 0; k < 3; k++) u[k] /= ul;
    v[0] = w[1] * u[2] - w[2] * u[1];
    v[1] = w[2] * u[0] - w[0] * u[2];
    v[2] = w[0] * u[1] - w[1] * u[0];
    for (int k = 0; k < 3; k++) {
        double hor = (double)focus * vw * u[k], ver = (double)focus * vh * v[k];
        c->origin[k] = from[k];
        c->horizontal[k] = (float)hor;
        c->vertical[k] = (float)ver;
        c->lower_left[k] = (float)((double)from[k] - hor / 2.0 - ver / 2.0 - (double)focus * w[k]);
        c->u[k] = (float)u[k];
        c->v[k] = (float)v[k];
        c->w[k] = (float)w[k];
    }
    c->lens_radius = (float)((double)aperture / 2.0);
    return RT_OK;
}

static void set_mat(RtMaterial* m, uint32_t kind, float r, float g, float b, float fuzz, float ior) {
    memset(m, 0, sizeof *m);
    m->kind = kind; m->albedo[0] = r; m->albedo[1] = g; m->albedo[2] = b;
    m->fuzz = fuzz; m->ior = ior;
}
static void set_sph(RtSphere* s, float x, float y, float z, float r) {
    s->cx = x; s->cy = y; s->cz = z; s->radius = r;
}

int oracle_make_cover_scene(uint32_t seed, int grid_half, RtSphere* sph, RtMaterial* mat,
                            uint32_t cap, uint32_t* out_n) {
    /* SURVEY 9.6 random_scene; RNG stream (seed, 0x5CE7E5EE, 0) */
    uint32_t st = rng_init(seed, 0x5CE7E5EEu, 0u);
    uint32_t n = 0;
    if (cap < 4) return RT_ERR_INVALID;
    set_sph(&sph[n], 0.0f, -1000.0f, 0.0f, 1000.0f);
    set_mat(&mat[n], RT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f);
    n++;
    for (int a = -grid_half; a < grid_half; a++) {
        for (int b = -grid_half; b < grid_half; b++) {
            float m = randf(&st);
            float cx = (float)a + 0.9f * randf(&st);
            float cz = (float)b + 0.9f * randf(&st);
            float dx = cx - 4.0f, dy = 0.2f - 0.2f, dz = cz - 0.0f;
            float len = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            if (!(len > 0.9f)) continue;
            if (n + 3 >= cap) return RT_ERR_INVALID;
            set_sph(&sph[n], cx, 0.2f, cz, 0.2f);
            if (m < 0.8f) {
                float a0 = randf(&st), a1 = randf(&st), a2 = randf(&st);
                float b0 = randf(&st), b1 = randf(&st), b2 = randf(&st);
                set_mat(&mat[n], RT_MAT_LAMBERTIAN, a0 * b0, a1 * b1, a2 * b2, 0.0f, 0.0f);
            } else if (m < 0.95f) {
                float a0 = 0.5f + 0.5f * randf(&st);
                float a1 = 0.5f + 0.5f * randf(&st);
                float a2 = 0.5f + 0.5f * randf(&st);
                float fz = 0.5f * randf(&st);
                set_mat(&mat[n], RT_MAT_METAL, a0, a1, a2, fz, 0.0f);
            } else {
                set_mat(&mat[n], RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 0.0f, 1.5f);
            }
            n++;
        }
    }
    set_sph(&sph[n], 0.0f, 1.0f, 0.0f, 1.0f);
    set_mat(&mat[n], RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 0.0f, 1.5f);
    n++;
    set_sph(&sph[n], -4.0f, 1.0f, 0.0f, 1.0f);
    set_mat(&mat[n], RT_MAT_LAMBERTIAN, 0.4f, 0.2f, 0.1f, 0.0f, 0.0f);
    n++;
    set_sph(&sph[n], 4.0f, 1.0f, 0.0f, 1.0f);
    set_mat(&mat[n], RT_MAT_METAL, 0.7f, 0.6f, 0.5f, 0.0f, 0.0f);
    n++;
    *out_n = n;
    return RT_OK;
}

int oracle_make_three_sphere_scene(int with_bubble, RtSphere* sph, RtMaterial* mat, uint32_t cap,
                                   uint32_t* out_n) {
    /* book ch.10 scene = BASELINE config 2 */
    uint32_t n = 0;
    if (cap < 5) return RT_ERR_INVALID;
    set_sph(&sph[n], 0.0f, -100.5f, -1.0f, 100.0f);
    set_mat(&mat[n++], RT_MAT_LAMBERTIAN, 0.8f, 0.8f, 0.0f, 0.0f, 0.0f);
    set_sph(&sph[n], 0.0f, 0.0f, -1.0f, 0.5f);
    set_mat(&mat[n++], RT_MAT_LAMBERTIAN, 0.1f, 0.2f, 0.5f, 0.0f, 0.0f);
    set_sph(&sph[n], -1.0f, 0.0f, -1.0f, 0.5f);
    set_mat(&mat[n++], RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 0.0f, 1.5f);
    if (with_bubble) {
        set_sph(&sph[n], -1.0f, 0.0f, -1.0f, -0.4f);
        set_mat(&mat[n++], RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 0.0f, 1.5f);
    }
    set_sph(&sph[n], 1.0f, 0.0f, -1.0f, 0.5f);
    set_mat(&mat[n++], RT_MAT_METAL, 0.8f, 0.6f, 0.2f, 0.0f, 0.0f);
    *out_n = n;
    return RT_OK;
}

int oracle_write_ppm(const char* path, const uint8_t* rgba8, uint32_t W, uint32_t H, size_t pitch) {
    /* P6, top scene row first = buffer row H-1 first (rt.frag:8) */
    FILE* f = fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    fprintf(f, "P6\n%u %u\n255\n", W, H);
    uint8_t* line = (uint8_t*)malloc((size_t)W * 3);
    if (!line) { fclose(f); return RT_ERR_NOMEM; }
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t* row = rgba8 + (size_t)(H - 1 - y) * pitch;
        for (uint32_t x = 0; x < W; x++) {
            line[3 * x + 0] = row[4 * x + 0];
            line[3 * x + 1] = row[4 * x + 1];
            line[3 * x + 2] = row[4 * x + 2];
        }
        fwrite(line, 1, (size_t)W * 3, f);
    }
    free(line);
    fclose(f);
    return RT_OK;
}

/* ---- arithmetic conformance probes (CPU side of tests/test_gpu_parity.py::test_arith_bit_exact) ---
 * op: 0 fma(a,b,c) 1 a/b 2 sqrt(a) 3 a*b 4 a+b 5 rng: 4th draw of stream (seed=a bits,
 * pixel=c bits, sample 7) 6 (float)(to_fixed(a)+to_fixed(b)) [fixed-point accumulate +
 * u64->float rounding] 7 (float)(u64 built from the bits of a (high) and b (low)) */
int oracle_arith(uint32_t op, const float* a, const float* b, const float* c, float* out,
                 uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        switch (op) {
            case 0: out[i] = fmaf(a[i], b[i], c[i]); break;
            case 1: out[i] = a[i] / b[i]; break;
            case 2: out[i] = sqrtf(a[i]); break;
            case 3: out[i] = a[i] * b[i]; break;
            case 4: out[i] = a[i] + b[i]; break;
            case 5: {
                uint32_t ua, uc;
                memcpy(&ua, &a[i], 4);
                memcpy(&uc, &c[i], 4);
                uint32_t st = rng_init(ua, uc, 7u);
                float r = 0;
                for (int k = 0; k < 4; k++) r = randf(&st);
                out[i] = r;
                break;
            }
            case 6: out[i] = (float)(to_fixed(a[i]) + to_fixed(b[i])); break;
            case 7: {
                uint32_t hi, lo;
                memcpy(&hi, &a[i], 4);
                memcpy(&lo, &b[i], 4);
                out[i] = (float)(((uint64_t)hi << 32) | lo);
                break;
            }
            default: return RT_ERR_INVALID;
        }
    }
    return RT_OK;
}

int oracle_cpu_ok(void) {
#if defined(__x86_64__) && defined(__FMA__)
    return __builtin_cpu_supports("fma") ? 1 : 0;
#else
    return 1;
#endif
}
