#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int iters, long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
    f4 c4[8];
    for (int i = 0; i < 8; ++i) c4[i] = (f4){0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0; for (int i = 0; i < 8; ++i) acc += c4[i][0];
    if (lane == 0 && blockIdx.x == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = r1 - r0; }
    if (acc == 12345.678f) sink[0] = acc;
}
int main() {
    long long* d_out; float* sink; long long h[16];
    hipMalloc(&d_out, 128); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 256}) {
        const int iters = 200000;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, 1000, d_out, sink);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, iters, d_out, sink);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d_out, 128, hipMemcpyDeviceToHost);
        const double mfma = 8.0 * iters * 2;     // per SIMD (2 waves)
        printf("blocks %d: %.3f ms; memtime ticks %lld (%.3f GHz), memrealtime %lld (%.1f MHz); per SIMD %.2f ns per MFMA = %.1f TFLOP/s/chip-equivalent(1024 SIMDs); ticks per MFMA (SIMD) %.2f\n",
               blocks, ms, h[0], h[0] / (ms * 1e6), h[1], h[1] / (ms * 1e3), ms * 1e6 / mfma, 16384.0 / (ms * 1e-3 / mfma) * 1024 / 1e12, h[0] / mfma);
    }
    return 0;
}
