#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long* out, int n) {
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(), w0 = wall_clock64(), c0 = clock64();
    float x = threadIdx.x;
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 0.999f, 0.5f);
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(), w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = w1 - w0; out[3] = c1 - c0; }
    if (x == 1234.5f) out[4] = 1;
}
int main() {
    int wc = 0, sc = 0;
    hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&sc, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate attr %d kHz, shader clock rate attr %d kHz\n", wc, sc);
    long long* d; hipMalloc(&d, 64); long long h[5];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 5000000);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    printf("events %.3f ms; s_memtime %lld (%.3f GHz); s_memrealtime %lld (%.2f MHz); wall_clock64 %lld (%.2f MHz); clock64 %lld (%.3f GHz); dependent fma: %.2f memtime ticks each\n",
           ms, h[0], h[0] / ms / 1e6, h[1], h[1] / ms / 1e3, h[2], h[2] / ms / 1e3, h[3], h[3] / ms / 1e6, h[0] / 5e6);
    return 0;
}
