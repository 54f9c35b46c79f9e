// Does gfx950 execute scalar memory atomics?  (tools/blockprof counts basic-block executions with one s_atomic_add per block.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p, int iters) {
    unsigned one = 1u;
    for (int i = 0; i < iters; ++i) {
        asm volatile("s_atomic_add %0, %1, 0x0\n\ts_atomic_add %0, %1, 0x40\n\ts_waitcnt lgkmcnt(0)" ::"s"(one), "s"(p) : "memory");
    }
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    const int blocks = 2048, threads = 256, iters = 100;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(b);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    unsigned h[32];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("sync %s; counter0 %u counter@0x40 %u expected %u; %.3f ms = %.1f atomics/us\n", hipGetErrorString(e), h[0], h[16], blocks * (threads / 64) * iters, ms,
           2.0 * blocks * (threads / 64) * iters / (ms * 1000.0));
    return h[0] == (unsigned)(blocks * (threads / 64) * iters) ? 0 : 1;
}
