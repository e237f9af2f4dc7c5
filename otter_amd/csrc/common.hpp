// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the OTTER hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

static constexpr int kWave = 64;   // CDNA wavefront width

// ---- error plumbing across the C ABI (no exceptions cross it) -------------------------------
enum OtterStatus {
    OTTER_OK = 0,
    OTTER_ERR_INVALID_ARGUMENT = 1,
    OTTER_ERR_UNSUPPORTED_SHAPE = 2,
    OTTER_ERR_LAUNCH = 3,
    OTTER_ERR_ALIGNMENT = 4,
};

void otter_set_error(const char* fmt, ...);   // defined in capi.hip (thread-local message)
int otter_reserve_lds(const void* kernel, size_t bytes, const char* name);   // capi.hip: per (kernel, device) dynamic-LDS opt-in; OTTER_OK or error set
int otter_cu_count(const char* name);                                       // capi.hip: CUs of the current device, 0 = error set

#define OTTER_REQUIRE(cond, code, ...)                 \
    do {                                               \
        if (!(cond)) {                                 \
            otter_set_error(__VA_ARGS__);              \
            return (code);                             \
        }                                              \
    } while (0)

#define OTTER_CHECK_LAUNCH(name)                                                    \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            otter_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return OTTER_ERR_LAUNCH;                                                \
        }                                                                           \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- element types ------------------------------------------------------------------------------
// Data type codes of the C ABI (OTTER_DT_* in otter_hip.h).
enum { DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2 };

// 16-bit MFMA operand types.  bf16 keeps fp32's range; fp16 carries 3 more mantissa bits at the same MFMA rate
// (CLIP's published weights are fp16).  Conversions to fp16 saturate instead of overflowing to infinity.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// MODE.FP16_OVFL (bit 23 of the wave's MODE register): a result that overflows fp16 -- arithmetic or a conversion from fp32 --
// is clamped to +-65504 instead of becoming an infinity (true infinities stay).  Set once at kernel entry, it makes every
// fp32 -> fp16 conversion of the kernel saturate without the two-instruction clamp in front of it.
__device__ __forceinline__ void fp16_saturate_on_overflow() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }

template <typename T> struct Half16;
template <> struct Half16<__bf16> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    typedef bf16x2 v2;
    // two fp32 -> one packed pair (bf16 keeps fp32's range: nothing to saturate)
    static __device__ __forceinline__ bf16x2 pack_sat_hw(float a, float b) { return (bf16x2){(__bf16)a, (__bf16)b}; }
    static __device__ __forceinline__ __bf16 from(float v) { return (__bf16)v; }
    static __device__ __forceinline__ __bf16 from_unit(float v) { return (__bf16)v; }      // v in [0, 1]
    static __device__ __forceinline__ f32x4 mfma32(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32x32(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
};
template <> struct Half16<_Float16> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    typedef f16x2 v2;
    // two fp32 -> one packed pair, round to nearest even; saturating only in a kernel that called fp16_saturate_on_overflow()
    static __device__ __forceinline__ f16x2 pack_sat_hw(float a, float b) { return (f16x2){(_Float16)a, (_Float16)b}; }
    static __device__ __forceinline__ _Float16 from(float v) { return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f); }
    static __device__ __forceinline__ _Float16 from_unit(float v) { return (_Float16)v; }  // v in [0, 1]: no saturation needed
    static __device__ __forceinline__ f32x4 mfma32(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32x32(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) {
        union { s16x4 s; f16x4 h; } ua, ub;
        ua.s = a; ub.s = b;
        return __builtin_amdgcn_mfma_f32_16x16x16f16(ua.h, ub.h, c, 0, 0, 0);
    }
};
template <typename T> __device__ __forceinline__ short bits16(T v) {
    union { T t; short s; } u;
    u.t = v;
    return u.s;
}

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f32_to_bf16(float v) { return (__bf16)v; }   // RNE, NaN-preserving (v_cvt_pk_bf16_f32)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// reductions over the four 16-lane rows of a wave (same lane index li in every row) with v_permlane{32,16}_swap: VALU only,
// no LDS round trip.  Every lane ends up with the result.
__device__ __forceinline__ float rows_max(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// (mean, 1 / sqrt(var + eps)) of one row of `width` 16-bit values held 8 per lane and vector: vector i of lane l is elements
// (i * 64 + l) * 8 .. + 7 of the row (zero beyond the width).  Two-pass on the registers; every lane gets the result.
// Shared by the row-statistics kernel and by the split-K reduce that finishes a residual row and takes its statistics.
template <typename T>
__device__ __forceinline__ f32x2 row_mean_rstd_16(const typename Half16<T>::v8 (&v)[2], int lane, int nvec, float inv_w, float eps) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += (float)v[i][e];
    const float mean = wave_sum(sum) * inv_w;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i * 64 + lane < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)v[i][e] - mean; sq += d * d; }
        }
    }
    return (f32x2){mean, rsqrtf(wave_sum(sq) * inv_w + eps)};
}

// x * sigmoid(1.702 x) with the hardware 2^x and reciprocal (1 ulp each; the IEEE division the plain expression compiles to
// costs 10 instructions per element, 8 % of a 256x256x1024 GEMM tile)
__device__ __forceinline__ float quick_gelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504088896341f * x));
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// The same function for epilogues whose pace the vector ALU sets (the erf-GELU tower MLPs of OpenCLIP checkpoints): x * Phi(x)
// with Phi(-|x|) = 0.5 erfc(|x| / sqrt 2) from Abramowitz & Stegun 7.1.26 (erfc(z) = poly(1 / (1 + p z)) e^(-z^2), absolute
// error 1.5e-7): 15 branch-free instructions, two of them transcendental, against ~45 with branches for the library erff.
// |gelu_erf_fast(x) - x Phi(x)| <= 2e-7 * max(1, |x|) (checked on a 1e-5 grid over [-10, 10] in fp32 arithmetic).
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float q = __builtin_fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    q = __builtin_fmaf(q, t, 0.5f * 1.421413741f);
    q = __builtin_fmaf(q, t, 0.5f * -0.284496736f);
    q = __builtin_fmaf(q, t, 0.5f * 0.254829592f);
    q = q * t * __builtin_amdgcn_exp2f(z * z * -1.44269504088896341f);      // 0.5 erfc(z) = Phi(-|x|)
    return x * (x < 0.f ? q : 1.0f - q);
}

// Bijective remap of a linear workgroup id so that each of the 8 XCDs (which receive workgroups
// round-robin) works on one contiguous chunk of the tile space and its private L2 sees neighbouring tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < nx) return bid;
    int q = nwg / nx, r = nwg % nx;
    int xcd = bid % nx, idx = bid / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
