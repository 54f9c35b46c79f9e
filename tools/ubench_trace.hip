// ubench_trace.hip — the ray-sphere fast loop of path_persistent_kernel in isolation:
// R path slots per lane, N spheres, sign bits shifted into candidate words.  Measures the
// ceiling (sphere tests / s) of the formulation on the GPU at hand, for the LDS-broadcast and
// the scalar-load (SGPR operand) variants, at several occupancies.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-fno-slp-vectorize] -o ubench_trace ubench_trace.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define DI __device__ __forceinline__
DI float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <int R, bool LDS, int WPS>
__global__ __launch_bounds__(256, WPS) void k_trace(const float4* __restrict__ sph, uint32_t n, int iters, uint32_t* out) {
    extern __shared__ float4 lds[];
    if (LDS) {
        for (uint32_t i = threadIdx.x; i < n; i += 256) lds[i] = sph[i];
        __syncthreads();
    }
    float ox[R], oy[R], oz[R], dx[R], dy[R], dz[R];
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ox[r] = 13.0f + 0.001f * (t % 97) + r; oy[r] = 2.0f + 0.002f * (t % 89); oz[r] = 3.0f - 0.001f * (t % 83);
        float a = -0.9f + 0.0001f * (t % 1013), b = -0.1f - 0.0002f * (t % 499), c = -0.2f + 0.0003f * (t % 251) + 0.01f * r;
        const float k = 1.0f / __builtin_sqrtf(a * a + b * b + c * c);
        dx[r] = a * k; dy[r] = b * k; dz[r] = c * k;
    }
    uint32_t check = 0;
    for (int it = 0; it < iters; ++it) {
        for (uint32_t base = 0; base < n; base += 32) {
            uint32_t miss[R];
#pragma unroll
            for (int r = 0; r < R; ++r) miss[r] = 0;
#pragma unroll 8
            for (uint32_t j = 0; j < 32; ++j) {
                const float4 s = LDS ? lds[base + j] : sph[base + j];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float ocx = ox[r] - s.x, ocy = oy[r] - s.y, ocz = oz[r] - s.z;
                    const float hb = fma_(ocz, dz[r], fma_(ocy, dy[r], ocx * dx[r]));
                    const float cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -s.w)));
                    const float disc = fma_(hb, hb, -cc);
                    miss[r] = __builtin_amdgcn_alignbit(miss[r], __float_as_uint(disc), 31);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) check += __builtin_popcount(~miss[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) ox[r] += 1e-4f;  // keep iterations from being hoisted
    }
    out[t] = check;
}

template <int R, bool LDS, int WPS>
static int run(const float4* d_sph, uint32_t n, uint32_t* d_out, int cus, const char* name) {
    const int iters = 64;
    const int grid = cus * WPS;  // WPS waves per SIMD = WPS blocks of 4 waves per CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_trace<R, LDS, WPS>), dim3(grid), dim3(256), LDS ? n * 16 : 0, 0, d_sph, n, iters, d_out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double tests = double(grid) * 256 * R * iters * n;
    printf("%-28s R=%d waves/SIMD=%d: %.3f ms  %.2f Ttests/s  %.1f TFLOP/s(16/test)\n", name, R, WPS, best,
           tests / best / 1e9, tests * 16 / best / 1e9);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const uint32_t n = 512;
    std::vector<float4> h(n);
    for (uint32_t i = 0; i < n; ++i) h[i] = make_float4(-11.f + (i % 22) + 0.3f, 0.2f, -11.f + (i / 22) + 0.6f, 0.04f);
    float4* d_sph; uint32_t* d_out;
    CHECK(hipMalloc(&d_sph, n * 16)); CHECK(hipMalloc(&d_out, sizeof(uint32_t) * 256 * cus * 8));
    CHECK(hipMemcpy(d_sph, h.data(), n * 16, hipMemcpyHostToDevice));
    if (run<1, true, 4>(d_sph, n, d_out, cus, "lds")) return 1;
    if (run<1, true, 8>(d_sph, n, d_out, cus, "lds")) return 1;
    if (run<2, true, 2>(d_sph, n, d_out, cus, "lds")) return 1;
    if (run<2, true, 4>(d_sph, n, d_out, cus, "lds")) return 1;
    if (run<2, true, 5>(d_sph, n, d_out, cus, "lds")) return 1;
    if (run<2, true, 8>(d_sph, n, d_out, cus, "lds")) return 1;
    if (run<4, true, 2>(d_sph, n, d_out, cus, "lds")) return 1;
    if (run<4, true, 4>(d_sph, n, d_out, cus, "lds")) return 1;
    if (run<1, false, 8>(d_sph, n, d_out, cus, "scalar-load")) return 1;
    if (run<2, false, 4>(d_sph, n, d_out, cus, "scalar-load")) return 1;
    if (run<2, false, 8>(d_sph, n, d_out, cus, "scalar-load")) return 1;
    if (run<4, false, 4>(d_sph, n, d_out, cus, "scalar-load")) return 1;
    return 0;
}
