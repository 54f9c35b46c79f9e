// ubench_valu.hip — measures, on the MI355X at hand, the issue rates the path-trace inner
// loop is priced against: v_fma_f32, v_pk_fma_f32, and wave-uniform (broadcast) LDS reads
// beside FMAs.  Standalone: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ __launch_bounds__(256) void k_pkfma(float* out, int iters, float s) {
    float2v a0 = {float(threadIdx.x), 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float2v sv = {s, s};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                         "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sv));
        }
    }
    float2v r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;
}

// FPR fmas per broadcast ds_read_b128 (wave-uniform address)
template <int FPR>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters, int n) {
    extern __shared__ float4 lds[];
    for (int i = threadIdx.x; i < n; i += 256) lds[i] = make_float4(i * 1e-6f, 1e-6f, 2e-6f, 3e-6f);
    __syncthreads();
    float a0 = threadIdx.x * 1e-3f, a1 = a0, a2 = a0, a3 = a0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int i = 0; i < n; ++i) {
            const float4 s = lds[i];
#pragma unroll
            for (int f = 0; f < FPR; f += 4) {
                a0 = __builtin_fmaf(a0, s.x, s.y);
                a1 = __builtin_fmaf(a1, s.y, s.z);
                a2 = __builtin_fmaf(a2, s.z, s.w);
                a3 = __builtin_fmaf(a3, s.w, s.x);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <class F>
static float time_ms(F&& launch, int reps = 5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 16));
    const int iters = 4096;
    for (int bpc : {1, 2, 4, 8}) {
        const int grid = cus * bpc;
        const double n_inst = double(grid) * 4 /*waves*/ * iters * 64.0;  // wave-instructions
        float ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); });
        printf("v_fma_f32    %d blocks/CU: %.3f ms  %.1f TFLOP/s  %.2f cyc/wave-inst/SIMD @2.4GHz\n", bpc, ms,
               n_inst * 64 * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n_inst / (cus * 4)));
        ms = time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); });
        printf("v_pk_fma_f32 %d blocks/CU: %.3f ms  %.1f TFLOP/s  %.2f cyc/wave-inst/SIMD @2.4GHz\n", bpc, ms,
               n_inst * 64 * 4 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n_inst / (cus * 4)));
    }
    const int n = 512, lit = 64;
    for (int bpc : {1, 2, 4, 8}) {
        const int grid = cus * bpc;
        const double reads = double(grid) * 4 * lit * n;  // wave-level ds_read_b128
        float ms = time_ms([&] { hipLaunchKernelGGL(k_lds<4>, dim3(grid), dim3(256), n * 16, 0, out, lit, n); });
        printf("lds bcast b128 + 4 fma  %d blocks/CU: %.3f ms  %.2f cyc/read/CU  %.1f TFLOP/s\n", bpc, ms,
               ms * 1e-3 * 2.4e9 / (reads / cus), reads * 4 * 128 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_lds<12>, dim3(grid), dim3(256), n * 16, 0, out, lit, n); });
        printf("lds bcast b128 + 12 fma %d blocks/CU: %.3f ms  %.2f cyc/read/CU  %.1f TFLOP/s\n", bpc, ms,
               ms * 1e-3 * 2.4e9 / (reads / cus), reads * 12 * 128 / ms / 1e9);
    }
    hipFree(out);
    return 0;
}
