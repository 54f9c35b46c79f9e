// Exhaustive comparison (every positive finite float) of short correctly-rounded-square-root candidates against the lean form the
// kernels use (lean_sqrt: v_sqrt_f32 + the residual check of its two neighbours, 10 instructions) and against sqrtf.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o sqrt_scan tools/sqrt_scan.hip && ./sqrt_scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define DI __device__ __forceinline__
DI float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DI float lean_sqrt(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = fma_(-s_dn, s, x), r_up = fma_(-s_up, s, x);
    const float r = (0.0f >= r_dn) ? s_dn : s;
    return (0.0f < r_up) ? s_up : r;
}
DI float cand_a(float x) {  // rsq, one Newton step on the root
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    return fma_(fma_(-g, g, x), h, g);
}
DI float cand_b(float x) {  // ... and a second one
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float s = fma_(fma_(-g, g, x), h, g);
    return fma_(fma_(-s, s, x), h, s);
}
DI float cand_c(float x) {  // Markstein: refine h and g together, then the residual step
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float e = fma_(-h, g, 0.5f);
    const float h1 = fma_(h, e, h), g1 = fma_(g, e, g);
    return fma_(fma_(-g1, g1, x), h1, g1);
}
DI float cand_d(float x) {  // v_sqrt_f32 and one residual step with h = 0.5 * rsq
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    return fma_(fma_(-s, s, x), h, s);
}
DI float cand_e(float x) {  // cand_a with the operand kept away from zero (x = 0 -> 0)
    const float y = __builtin_amdgcn_rsqf(__builtin_fmaxf(x, 0x1p-126f));
    const float g = x * y, h = 0.5f * y;
    return fma_(fma_(-g, g, x), h, g);
}
DI float lean_div(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float r1 = fma_(fma_(-b, r0, 1.0f), r0, r0);
    const float q0 = a * r1;
    const float q1 = fma_(fma_(-b, q0, a), r1, q0);
    return fma_(fma_(-b, q1, a), r1, q1);
}
// k = RN(1 / RN(sqrt(x))) -- unit3_scattered's factor -- from the same v_rsq_f32: the exact root, then Newton steps on the reciprocal
DI void inv_root_cands(float x, float out[4]) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float s = fma_(fma_(-g, g, x), h, g);
    const float r1 = fma_(fma_(-s, y, 1.0f), y, y);
    const float q1 = fma_(fma_(-s, r1, 1.0f), r1, r1);
    const float q2 = fma_(fma_(-s, q1, 1.0f), r1, q1);
    const float r1b = fma_(fma_(-s, y, 1.0f), y, y);
    const float q1b = fma_(fma_(-s, r1b, 1.0f), y, r1b);  // second step with y instead of r1 as the slope
    out[0] = r1; out[1] = q1; out[2] = q2; out[3] = q1b;
}
__global__ void scan_inv(uint32_t lo, unsigned long long n, unsigned long long* bad, uint32_t* first) {
    const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float(lo + static_cast<uint32_t>(i));
    const float want = 1.0f / __builtin_sqrtf(x), lean = lean_div(1.0f, lean_sqrt(x));
    float got[5];
    inv_root_cands(x, got);
    got[4] = lean;
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (__float_as_uint(got[k]) != __float_as_uint(want)) {
            const unsigned long long at = atomicAdd(&bad[k], 1ull);
            if (at < 4) first[k * 4 + at] = __float_as_uint(x);
        }
}
// RN(1 / s) for a float s: v_rcp_f32 and Newton steps (the compiler's core, lean_div(1, s), takes 8 instructions)
__global__ void scan_rcp(uint32_t lo, unsigned long long n, unsigned long long* bad, uint32_t* first) {
    const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const float s = __uint_as_float(lo + static_cast<uint32_t>(i));
    const float want = 1.0f / s;
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float r1 = fma_(fma_(-s, r0, 1.0f), r0, r0);
    const float r2 = fma_(fma_(-s, r1, 1.0f), r1, r1);
    const float r2b = fma_(fma_(-s, r1, 1.0f), r0, r1);
    const float got[5] = {r0, r1, r2, r2b, lean_div(1.0f, s)};
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (__float_as_uint(got[k]) != __float_as_uint(want)) {
            const unsigned long long at = atomicAdd(&bad[k], 1ull);
            if (at < 4) first[k * 4 + at] = __float_as_uint(s);
        }
}
__global__ void scan(uint32_t lo, unsigned long long n, unsigned long long* bad, uint32_t* first) {
    const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float(lo + static_cast<uint32_t>(i));
    const float want = __builtin_sqrtf(x);
    const float got[6] = {lean_sqrt(x), cand_a(x), cand_b(x), cand_c(x), cand_d(x), cand_e(x)};
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (__float_as_uint(got[k]) != __float_as_uint(want)) {
            const unsigned long long at = atomicAdd(&bad[k], 1ull);
            if (at < 4) first[k * 4 + at] = __float_as_uint(x);
        }
}
int main() {
    unsigned long long* bad; uint32_t* first;
    hipMallocManaged(&bad, 6 * 8); hipMallocManaged(&first, 24 * 4);
    const char* names[6] = {"lean_sqrt", "rsq+1 step", "rsq+2 steps", "markstein", "sqrt+1 step", "rsq(max)+1 step"};
    // ranges: [2^-96, max finite] (the lean forms' domain), and the rest of the normals below
    const uint32_t edges[4] = {0x00800000u, 0x0F800000u /* 2^-96 */, 0x7F800000u, 0};
    for (int r = 0; r < 2; ++r) {
        for (int k = 0; k < 6; ++k) bad[k] = 0;
        for (int k = 0; k < 24; ++k) first[k] = 0;
        const unsigned long long n = edges[r + 1] - edges[r];
        hipLaunchKernelGGL(scan, dim3((n + 255) / 256), dim3(256), 0, 0, edges[r], n, bad, first);
        hipDeviceSynchronize();
        printf("floats [%08x, %08x): %llu\n", edges[r], edges[r + 1], n);
        for (int k = 0; k < 6; ++k) printf("  %-18s mismatches vs sqrtf: %llu   first: %08x %08x %08x %08x\n", names[k], bad[k], first[4*k], first[4*k+1], first[4*k+2], first[4*k+3]);
    }
    {   // the reciprocal root over unit3_scattered's domain and well beyond: [2^-64, 2^64]
        const char* inames[5] = {"root + 1 step", "root + 2 steps", "root + 3 steps", "root + 2 steps (slope y)", "lean_div(1, lean_sqrt)"};
        for (int k = 0; k < 6; ++k) bad[k] = 0;
        for (int k = 0; k < 24; ++k) first[k] = 0;
        const uint32_t lo = 0x1F800000u /* 2^-64 */, hi = 0x5F800000u /* 2^64 */;
        const unsigned long long n = hi - lo;
        hipLaunchKernelGGL(scan_inv, dim3((n + 255) / 256), dim3(256), 0, 0, lo, n, bad, first);
        hipDeviceSynchronize();
        printf("1 / sqrt(x), floats [2^-64, 2^64): %llu\n", n);
        for (int k = 0; k < 5; ++k) printf("  %-26s mismatches vs 1.0f / sqrtf(x): %llu   first: %08x %08x\n", inames[k], bad[k], first[4*k], first[4*k+1]);
    }
    {   // the reciprocal of a float in [2^-64, 2^64)
        const char* rnames[5] = {"v_rcp_f32", "rcp + 1 step", "rcp + 2 steps", "rcp + 2 steps (slope r0)", "lean_div(1, s)"};
        for (int k = 0; k < 6; ++k) bad[k] = 0;
        for (int k = 0; k < 24; ++k) first[k] = 0;
        const uint32_t lo = 0x1F800000u, hi = 0x5F800000u;
        const unsigned long long n = hi - lo;
        hipLaunchKernelGGL(scan_rcp, dim3((n + 255) / 256), dim3(256), 0, 0, lo, n, bad, first);
        hipDeviceSynchronize();
        printf("1 / s, floats [2^-64, 2^64): %llu\n", n);
        for (int k = 0; k < 5; ++k) printf("  %-26s mismatches vs 1.0f / s: %llu   first: %08x %08x\n", rnames[k], bad[k], first[4*k], first[4*k+1]);
    }
    return 0;
}
