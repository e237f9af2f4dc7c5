// Microbenchmark: how much vector-ALU work runs beside MFMAs of two shapes on one SIMD (two waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// mode bit0: MFMA waves active, bit1: VALU waves active; shape 0: 16x16x32, 1: 32x32x16; vkind 0: fma only, 1: exp only, 2: mix (1 exp : 2 fma)
template <int SHAPE, int VKIND>
__global__ __launch_bounds__(512) void k(int mode, int iters, long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool is_mfma = wave < 4;
    long long t0 = 0, t1 = 0;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
    float acc_out = 0.f;
    __syncthreads();
    if (is_mfma) {
        if (mode & 1) {
            if constexpr (SHAPE == 0) {
                f4 c[8];
                for (int i = 0; i < 8; ++i) c[i] = (f4){0, 0, 0, 0};
                t0 = __builtin_amdgcn_s_memtime();
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
                }
                t1 = __builtin_amdgcn_s_memtime();
                for (int i = 0; i < 8; ++i) acc_out += c[i][0];
            } else {
                f16v c[4];
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
                t0 = __builtin_amdgcn_s_memtime();
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
                }
                t1 = __builtin_amdgcn_s_memtime();
                for (int i = 0; i < 4; ++i) acc_out += c[i][0];
            }
        }
    } else {
        if (mode & 2) {
            float x[8];
            for (int i = 0; i < 8; ++i) x[i] = lane * 0.01f + i;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (VKIND == 0) { x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f); x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f); x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f); x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f); }
                    else if (VKIND == 1) { x[i] = __builtin_amdgcn_exp2f(x[i]); }
                    else { x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], 0.5f, -1.0f)); x[i] = fmaxf(x[i], 0.25f); }
                }
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int i = 0; i < 8; ++i) acc_out += x[i];
        }
    }
    if (lane == 0) out[wave] = t1 - t0;
    if (acc_out == 12345.678f) sink[0] = acc_out;
}

template <int SHAPE, int VKIND>
void run(const char* name, long long* d_out, float* sink) {
    const int iters = 2000;
    long long h[8];
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL((k<SHAPE, VKIND>), dim3(1), dim3(512), 0, 0, mode, iters, d_out, sink);
        hipLaunchKernelGGL((k<SHAPE, VKIND>), dim3(1), dim3(512), 0, 0, mode, iters, d_out, sink);
        hipDeviceSynchronize();
        hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-28s mode %d: mfma wave cycles/iter %.1f  valu wave cycles/iter %.1f\n", name, mode, (double)h[0] / iters, (double)h[4] / iters);
    }
}

// one wave per SIMD, MFMA and independent VALU interleaved in the same wave
template <int SHAPE>
__global__ __launch_bounds__(256) void k_same(int with_valu, int iters, long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane * 0.01f + i;
    float acc_out = 0.f;
    long long t0, t1;
    if constexpr (SHAPE == 0) {
        f4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = (f4){0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
                if (with_valu) { x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], 0.5f, -1.0f)); }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) acc_out += c[i][0];
    } else {
        f16v c[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
                if (with_valu) { x[2 * i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[2 * i], 0.5f, -1.0f)); x[2 * i + 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[2 * i + 1], 0.5f, -1.0f)); }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 4; ++i) acc_out += c[i][0];
    }
    for (int i = 0; i < 8; ++i) acc_out += x[i];
    if (lane == 0) out[wave] = t1 - t0;
    if (acc_out == 12345.678f) sink[0] = acc_out;
}

int main() {
    long long* d_out; float* sink;
    hipMalloc(&d_out, 64); hipMalloc(&sink, 4);
    printf("per iteration: 8 x 16x16x32 (=128 matrix cycles) or 4 x 32x32x16 (=128 matrix cycles); valu wave per iter: fma 32 instr, exp 8 instr, mix 8 exp + 8 fma + 8 max\n");
    run<0, 0>("16x16x32 + fma", d_out, sink);
    run<1, 0>("32x32x16 + fma", d_out, sink);
    run<0, 1>("16x16x32 + exp", d_out, sink);
    run<1, 1>("32x32x16 + exp", d_out, sink);
    run<0, 2>("16x16x32 + mix", d_out, sink);
    run<1, 2>("32x32x16 + mix", d_out, sink);
    long long h[4];
    for (int wv = 0; wv < 2; ++wv) {
        hipLaunchKernelGGL((k_same<0>), dim3(1), dim3(256), 0, 0, wv, 2000, d_out, sink); hipDeviceSynchronize();
        hipMemcpy(h, d_out, 32, hipMemcpyDeviceToHost);
        printf("same wave 16x16x32 with_valu=%d (8 mfma + 8 (fma+exp) per iter): %.1f cycles/iter\n", wv, h[0] / 2000.0);
        hipLaunchKernelGGL((k_same<1>), dim3(1), dim3(256), 0, 0, wv, 2000, d_out, sink); hipDeviceSynchronize();
        hipMemcpy(h, d_out, 32, hipMemcpyDeviceToHost);
        printf("same wave 32x32x16 with_valu=%d (4 mfma + 8 (fma+exp) per iter): %.1f cycles/iter\n", wv, h[0] / 2000.0);
    }
    return 0;
}
