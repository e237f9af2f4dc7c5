// 64x64-tile MFMA GEMM for small problems (same contract as gemm.hip; bf16 / fp16 / fp32 operands):
//   * the < 256 remainder rows of a big 16-bit GEMM whose whole 256-row tiles run in the 256x256 kernel (running
//     them beside the big kernel on a second stream was measured and lost to the cross-stream fork/join cost);
//   * any GEMM too small to give the 128x128 kernel one tile per CU (policy transformer, pooling projections, heads).
// Such problems are latency-bound: small tiles spread them over 4x more CUs and five LDS stages (two workgroups per
// CU) keep three K-steps of loads in flight per workgroup behind the one whose fragments are being read.
//
// Layout as in gemm.hip: LDS-DMA staging with the chunk swizzle on the source address, weight rows as the MFMA
// "A" operand; the weight rows of a wave's 32 columns are permuted so that a lane's two n-tiles hold 8
// consecutive output columns (16-byte stores), as in gemm256.hip.
#include <type_traits>

#include "common.hpp"
#include "otter_hip.h"

namespace {

constexpr int TB = 64;                                   // tile edge
constexpr int ROWB = 128;                                // bytes of K per row and K-step
constexpr int STAGE = 2 * TB * ROWB;                     // 16 KiB
// LDS stages NS (template parameter): NS-1 K-steps of loads in flight per workgroup.

struct TailArgs {
    const char* A; long lda;
    const char* W;
    const float* bias;
    const void* residual; long ldr;     // in the OUTPUT's type
    void* C; long ldc;
    int M, N, K, tiles_m, tiles_n;
    const float* ln_stats;      // [M][2] (mean, rstd) or null: LayerNorm folded into the epilogue (see otter_gemm_ln)
    const float* ln_colsum;     // [N]
    int k_splits;
    float* partial;             // split-K: [gridDim.y][M][N] fp32 partial products (the epilogue runs in splitk_reduce), or null
};

enum { ACT_NONE = 0, ACT_QUICKGELU = 1, ACT_GELU = 2, ACT_RELU = 3 };

template <int ACT> __device__ __forceinline__ float act_t(float v) {
    if (ACT == ACT_QUICKGELU) return quick_gelu(v);
    if (ACT == ACT_GELU) return gelu_erf(v);
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

template <typename T> struct TFrag { typedef typename Half16<T>::v8 type; };
template <> struct TFrag<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ f32x4 mma_t(const typename TFrag<T>::type& w, const typename TFrag<T>::type& a, f32x4 c) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], a[e], c, 0, 0, 0);
        return c;
    } else {
        return Half16<T>::mfma32(w, a, c);
    }
}

template <typename T, int ACT, bool OUT_F32, bool RES, int NS>
__global__ __launch_bounds__(256) void gemm_64x64(TailArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename TFrag<T>::type frag_t;
    constexpr int ES = sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    // Workgroup b runs on XCD b % 8, each with an L2 of its own: XCD x takes a contiguous run of the tile order (row tiles
    // fastest), i.e. all row tiles of a few column panels, so a weight panel is fetched into one L2 instead of all eight
    // (with the plain order every XCD pulled every panel of both operands through the fabric: 8 x 12.6 MB for 514 x 1024 x
    // 4096, which set the time of such launches).
    const int nt = p.tiles_m * p.tiles_n, xcd = blockIdx.x & 7;
    const int tile = xcd * (nt >> 3) + min(xcd, nt & 7) + (int)(blockIdx.x >> 3);
    const int tm = tile % p.tiles_m, tn = tile / p.tiles_m;
    const int m0 = tm * TB, n0 = tn * TB;

    // staging: a stage is 128 image rows (64 of A, then 64 of W) = 16 instructions of 8 rows; wave w issues 4w..4w+3
    // (waves 0,1 fetch A, waves 2,3 fetch W)
    unsigned src_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ir = (wave * 4 + i) * 8 + (lane >> 3);           // image row 0..127
        const int r = ir & 63;
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        if (wave < 2) {
            src_off[i] = (unsigned)min(m0 + r, p.M - 1) * (unsigned)(p.lda * ES) + c * 16;
        } else {
            const int fi = r & 15, fnt = (r >> 4) & 1;
            const int col = (r >> 5) * 32 + 8 * (fi >> 2) + 4 * fnt + (fi & 3);
            src_off[i] = (unsigned)(n0 + col) * (unsigned)(p.K * ES) + c * 16;
        }
    }
    const char* src_base = wave < 2 ? p.A : p.W;
    auto stage = [&](int buf, unsigned kbyte) {
        char* dst = smem + buf * STAGE + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src_base + (src_off[i] + kbyte)), LDS_PTR(dst + i * 1024), 16, 0, 0);
    };

    const int swz = (li >> 1) & 7;
    const int row_off = li * ROWB;

    f32x4 acc[2][2];                                    // [m-tile][n-tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // split-K: workgroup (x, y) multiplies K-steps [y*nk, (y+1)*nk) of the tile
    const int nk = p.K * ES / ROWB / (int)gridDim.y;
    const unsigned kbase = blockIdx.y * (unsigned)nk * ROWB;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) stage(t, kbase + (unsigned)t * ROWB);

    // The K loop is a chain per wave -- wait for a stage, barrier, fragment reads (LDS latency), 8 MFMAs, 4 LDS-DMA
    // requests (which hold the wave while the CU's address unit works through the four waves' 16 KiB) -- and the four waves
    // run it in lock-step, so nothing of it overlaps unless it is made to: the fragments of stage t+1 are read one
    // iteration ahead into a second register set (their latency passes under the MFMAs of stage t), and the requests of
    // stage t+NS-1 follow the MFMAs (which run in the matrix pipe meanwhile).  Stage t+NS-1 overwrites the buffer of stage
    // t-1, whose fragments every wave had in registers before it issued MFMAs(t-1), i.e. before it reached barrier t.
    frag_t wf[2][2][2], af[2][2][2];                    // [register set][k-half][tile]
    auto read_frags = [&](auto set_tag, int t) {
        constexpr int S = decltype(set_tag)::value;
        const int cur = t % NS;
        const char* sA = smem + cur * STAGE + wm * 32 * ROWB;
        const char* sW = smem + cur * STAGE + TB * ROWB + wn * 32 * ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = row_off + (((ks * 4 + g) ^ swz) << 4);
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[S][ks][j] = *(const frag_t*)(sW + j * 16 * ROWB + off);
#pragma unroll
            for (int i = 0; i < 2; ++i) af[S][ks][i] = *(const frag_t*)(sA + i * 16 * ROWB + off);
        }
    };
    // stage 0: min(nk, NS-1) stages are in flight
    if (NS - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    read_frags(std::integral_constant<int, 0>{}, 0);
    auto k_step = [&](auto set_tag, int t) {
        constexpr int S = decltype(set_tag)::value;
        if (t + 1 < nk) {
            // stages t+1 .. min(nk-1, t+NS-2) are in flight: retire stage t+1 only
            if (t + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 3)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");       // stage t+1 visible to all waves; everyone holds stage t in registers
        }
        // MFMAs first: the wait for their fragments (requested an iteration ago) must not cover the reads issued below
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma_t<T>(wf[S][ks][j], af[S][ks][i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nk) read_frags(std::integral_constant<int, S ^ 1>{}, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + NS - 1 < nk) stage((t + NS - 1) % NS, kbase + (unsigned)(t + NS - 1) * ROWB);
    };
    {
        int t = 0;
        for (; t + 1 < nk; t += 2) {
            k_step(std::integral_constant<int, 0>{}, t);
            k_step(std::integral_constant<int, 1>{}, t + 1);
        }
        if (t < nk) k_step(std::integral_constant<int, 0>{}, t);
    }

    // epilogue: lane holds row m = li (per m-tile) and columns 8g + 0..7 of the wave's 32
    const int n = n0 + wn * 32 + g * 8;
    if (n >= p.N) return;
    if (p.partial) {
        float* part = p.partial + (long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 32 + i * 16 + li;
            if (m >= p.M) continue;
            *(f32x4*)(part + (long)m * p.N + n) = acc[i][0];
            *(f32x4*)(part + (long)m * p.N + n + 4) = acc[i][1];
        }
        return;
    }
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        b0 = *(const f32x4*)(p.bias + n);
        b1 = *(const f32x4*)(p.bias + n + 4);
    }
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    if (p.ln_stats) {
        c0 = *(const f32x4*)(p.ln_colsum + n);
        c1 = *(const f32x4*)(p.ln_colsum + n + 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + li;
        if (m >= p.M) continue;
        f32x4 v0 = acc[i][0], v1 = acc[i][1];
        if (p.ln_stats) {
            const f32x2 st = *(const f32x2*)(p.ln_stats + 2 * (long)m);
            v0 = (v0 - st[0] * c0) * st[1];
            v1 = (v1 - st[0] * c1) * st[1];
        }
        v0 += b0;
        v1 += b1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] = act_t<ACT>(v0[e]);
            v1[e] = act_t<ACT>(v1[e]);
        }
        if (RES) {
            if constexpr (OUT_F32 || sizeof(T) == 4) {
                const float* r = (const float*)p.residual + (long)m * p.ldr + n;
                v0 += *(const f32x4*)r;
                v1 += *(const f32x4*)(r + 4);
            } else {
                const typename Half16<T>::v8 r = *(const typename Half16<T>::v8*)((const T*)p.residual + (long)m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] += (float)r[e];
                    v1[e] += (float)r[4 + e];
                }
            }
        }
        if constexpr (OUT_F32 || sizeof(T) == 4) {
            float* c = (float*)p.C + (long)m * p.ldc + n;
            *(f32x4*)c = v0;
            *(f32x4*)(c + 4) = v1;
        } else {
            typename Half16<T>::v8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = Half16<T>::from(v0[e]);
                o[4 + e] = Half16<T>::from(v1[e]);
            }
            *(typename Half16<T>::v8*)((T*)p.C + (long)m * p.ldc + n) = o;
        }
    }
}

template <typename T, int ACT, bool OUT_F32, bool RES, int NS>
int launch_tail_ns(const TailArgs& a, hipStream_t s) {
    auto kern = gemm_64x64<T, ACT, OUT_F32, RES, NS>;
    if (int rc = otter_reserve_lds((const void*)kern, NS * STAGE, "otter_gemm (64x64)")) return rc;
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n, a.partial ? a.k_splits : 1), dim3(256), NS * STAGE, s, a);
    OTTER_CHECK_LAUNCH("otter_gemm (64x64)");
    return OTTER_OK;
}

template <typename T, int ACT, bool OUT_F32, bool RES>
int launch_tail(const TailArgs& a, hipStream_t s) {
    // (8 stages for launches with fewer tiles than CUs were measured: no change -- such launches are bound by the 16 KiB
    //  per K-step that one CU can pull through its L1, not by the pipeline depth)
    return launch_tail_ns<T, ACT, OUT_F32, RES, 5>(a, s);
}

template <int ACT>
int launch_tail_f32(const TailArgs& a, hipStream_t s) { return launch_tail<float, ACT, true, false>(a, s); }

int dispatch_tail_f32(const TailArgs& a, int act, bool res, hipStream_t s) {
    if (res) return launch_tail<float, ACT_NONE, true, true>(a, s);
    switch (act) {
        case OTTER_EPI_ACT_NONE: return launch_tail_f32<ACT_NONE>(a, s);
        case OTTER_EPI_ACT_QUICKGELU: return launch_tail_f32<ACT_QUICKGELU>(a, s);
        case OTTER_EPI_ACT_GELU: return launch_tail_f32<ACT_GELU>(a, s);
        case OTTER_EPI_ACT_RELU: return launch_tail_f32<ACT_RELU>(a, s);
    }
    otter_set_error("otter_gemm: unknown activation %d", act);
    return OTTER_ERR_INVALID_ARGUMENT;
}

template <typename T>
int dispatch_tail(const TailArgs& a, int act, bool out_f32, bool res, hipStream_t s) {
    if (res) return out_f32 ? launch_tail<T, ACT_NONE, true, true>(a, s) : launch_tail<T, ACT_NONE, false, true>(a, s);
    switch (act) {
        case OTTER_EPI_ACT_NONE: return out_f32 ? launch_tail<T, ACT_NONE, true, false>(a, s) : launch_tail<T, ACT_NONE, false, false>(a, s);
        case OTTER_EPI_ACT_QUICKGELU: return out_f32 ? launch_tail<T, ACT_QUICKGELU, true, false>(a, s) : launch_tail<T, ACT_QUICKGELU, false, false>(a, s);
        case OTTER_EPI_ACT_GELU: return out_f32 ? launch_tail<T, ACT_GELU, true, false>(a, s) : launch_tail<T, ACT_GELU, false, false>(a, s);
        case OTTER_EPI_ACT_RELU: return out_f32 ? launch_tail<T, ACT_RELU, true, false>(a, s) : launch_tail<T, ACT_RELU, false, false>(a, s);
    }
    otter_set_error("otter_gemm: unknown activation %d", act);
    return OTTER_ERR_INVALID_ARGUMENT;
}

// Second pass of a split-K product: sums the partial products and applies the epilogue of the one-pass kernels
// (folded LayerNorm, bias, activation, residual in the output's type).  One thread per 4 consecutive columns.
struct ReduceArgs {
    const float* partial; int splits;
    const float* bias; const void* residual; long ldr;
    void* C; long ldc;
    int M, N, act, out_dt, res;
    const float* ln_stats; const float* ln_colsum;
};

template <int SP>
__device__ __forceinline__ f32x4 sum_partials(const float* part, long stride) {
    f32x4 pv[SP];
#pragma unroll
    for (int s = 0; s < SP; ++s) pv[s] = *(const f32x4*)(part + s * stride);
    f32x4 v = pv[0];
#pragma unroll
    for (int s = 1; s < SP; ++s) v += pv[s];
    return v;
}

__global__ __launch_bounds__(256) void splitk_reduce(ReduceArgs p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = p.N >> 2;
    if (i >= (long)p.M * nq) return;
    const int m = (int)(i / nq), n = (int)(i % nq) * 4;
    // all partial products in flight at once (a loop with a run-time bound waits out one memory round trip per split),
    // summed in split order
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* part = p.partial + (long)m * p.N + n;
    const long stride = (long)p.M * p.N;
    if (p.splits == 8) v = sum_partials<8>(part, stride);
    else if (p.splits == 4) v = sum_partials<4>(part, stride);
    else if (p.splits == 2) v = sum_partials<2>(part, stride);
    else
        for (int s = 0; s < p.splits; ++s) v += *(const f32x4*)(part + s * stride);
    if (p.ln_stats) {
        const f32x2 st = *(const f32x2*)(p.ln_stats + 2 * (long)m);
        v = (v - st[0] * *(const f32x4*)(p.ln_colsum + n)) * st[1];
    }
    if (p.bias) v += *(const f32x4*)(p.bias + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (p.act == ACT_QUICKGELU) v[e] = quick_gelu(v[e]);
        else if (p.act == ACT_GELU) v[e] = gelu_erf(v[e]);
        else if (p.act == ACT_RELU) v[e] = fmaxf(v[e], 0.f);
    }
    if (p.out_dt == DT_F32) {
        if (p.res) v += *(const f32x4*)((const float*)p.residual + (long)m * p.ldr + n);
        *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = v;
    } else if (p.out_dt == DT_F16) {
        if (p.res) {
            const f16x4 r = *(const f16x4*)((const _Float16*)p.residual + (long)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
        }
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = Half16<_Float16>::from(v[e]);
        *(f16x4*)((_Float16*)p.C + (long)m * p.ldc + n) = o;
    } else {
        if (p.res) {
            const bf16x4 r = *(const bf16x4*)((const __bf16*)p.residual + (long)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = Half16<__bf16>::from(v[e]);
        *(bf16x4*)((__bf16*)p.C + (long)m * p.ldc + n) = o;
    }
}

// Second pass of a split-K product whose output is a 16-bit residual stream (x += A W^T + bias on the last `rem` rows of x),
// fused with the row statistics of the WHOLE stream: the launch that would only sum the partial products of a few rows rides
// on the one that reads every row for the LayerNorm folded into the next GEMM.  One wave per row, 8 columns per lane and
// vector.  The first blocks take the `rem` product rows (partial sums in split order, bias, residual, rounding: the
// arithmetic of splitk_reduce; statistics of the rounded row: the arithmetic of the row-statistics kernel), the others the
// `m_main` rows in front of them, which the big-tile kernel has already finished.
struct ReduceStatsArgs {
    const float* partial; int splits;       // [splits][rem][N]
    const float* bias;
    void* x; long ldx;                      // the stream: rows [0, m_main + rem), 16-bit
    float* stats;                           // [m_main + rem][2]
    int m_main, rem, N;
    float eps;
};

template <typename T, int SP>
__global__ __launch_bounds__(256) void reduce_rows_stats(ReduceStatsArgs p) {
    typedef typename Half16<T>::v8 v8;
    const int lane = threadIdx.x & 63;
    const int rem_blocks = (p.rem + 3) >> 2;
    const int nvec = p.N >> 3;
    const float inv_w = 1.0f / (float)p.N;
    v8 v[2];
    int row;
    if ((int)blockIdx.x < rem_blocks) {
        const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (r >= p.rem) return;
        row = p.m_main + r;
        T* xr = (T*)p.x + (long)row * p.ldx;
        const long stride = (long)p.rem * p.N;
        // one 4-column group at a time (its SP partial sums in flight together): the register count of this branch is the
        // occupancy of the whole launch, and the other branch streams the stream at the memory system's rate
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = i * 64 + lane;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = Half16<T>::from(0.f);
            if (idx < nvec) {
                const float* part = p.partial + (long)r * p.N + idx * 8;
                const v8 res = *(const v8*)(xr + idx * 8);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 pv[SP];
#pragma unroll
                    for (int s = 0; s < SP; ++s) pv[s] = *(const f32x4*)(part + s * stride + 4 * h);
                    f32x4 a = pv[0];
#pragma unroll
                    for (int s = 1; s < SP; ++s) a += pv[s];
                    if (p.bias) a += *(const f32x4*)(p.bias + idx * 8 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][4 * h + e] = Half16<T>::from(a[e] + (float)res[4 * h + e]);
                    asm volatile("" ::: "memory");          // keeps the next group's loads behind this one's arithmetic
                }
                *(v8*)(xr + idx * 8) = v[i];
            }
        }
    } else {
        row = ((int)blockIdx.x - rem_blocks) * 4 + (threadIdx.x >> 6);
        if (row >= p.m_main) return;
        const T* xr = (const T*)p.x + (long)row * p.ldx;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = i * 64 + lane;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = Half16<T>::from(0.f);
            if (idx < nvec) v[i] = *(const v8*)(xr + idx * 8);
        }
    }
    const f32x2 st = row_mean_rstd_16<T>(v, lane, nvec, inv_w, p.eps);
    if (lane == 0) *(f32x2*)(p.stats + 2 * (long)row) = st;
}

template <typename T>
int launch_reduce_rows_stats(const ReduceStatsArgs& r, hipStream_t s) {
    const unsigned grid = (unsigned)((r.rem + 3) / 4 + (r.m_main + 3) / 4);
    switch (r.splits) {
        case 2: hipLaunchKernelGGL((reduce_rows_stats<T, 2>), dim3(grid), dim3(256), 0, s, r); break;
        case 4: hipLaunchKernelGGL((reduce_rows_stats<T, 4>), dim3(grid), dim3(256), 0, s, r); break;
        case 8: hipLaunchKernelGGL((reduce_rows_stats<T, 8>), dim3(grid), dim3(256), 0, s, r); break;
        default: otter_set_error("otter_gemm_residual_stats: %d splits", r.splits); return OTTER_ERR_UNSUPPORTED_SHAPE;
    }
    OTTER_CHECK_LAUNCH("otter_gemm_residual_stats (split-K reduce + row statistics)");
    return OTTER_OK;
}

}  // namespace

// Split-K for problems with too few 64x64 tiles to occupy the chip (the < 256 remainder rows of the tower GEMMs: 32
// tiles, each pulling 1 MB through one CU): `splits` workgroups per tile write fp32 partial products to `workspace`
// ([splits][M][N] floats, caller-provided), a second launch sums them and applies the epilogue.
int otter_gemm64_splitk_launch(int dtype, const void* A, long lda, const void* W, const float* bias, const void* residual, long ldr,
                               void* C, long ldc, int M, int N, int K, int act, bool out_f32, bool res, const float* ln_stats,
                               const float* ln_colsum, float* workspace, int splits, hipStream_t s) {
    TailArgs a{(const char*)A, lda, (const char*)W, nullptr, nullptr, 0, nullptr, 0, M, N, K, (M + TB - 1) / TB, (N + TB - 1) / TB,
               nullptr, nullptr, splits, workspace};
    int rc = dtype == DT_BF16 ? launch_tail_ns<__bf16, ACT_NONE, true, false, 5>(a, s)
             : dtype == DT_F16 ? launch_tail_ns<_Float16, ACT_NONE, true, false, 5>(a, s)
                               : launch_tail_ns<float, ACT_NONE, true, false, 5>(a, s);
    if (rc != OTTER_OK) return rc;
    ReduceArgs r{workspace, splits, bias, residual, ldr, C, ldc, M, N, act, out_f32 ? DT_F32 : dtype, res ? 1 : 0, ln_stats, ln_colsum};
    const long threads = (long)M * (N / 4);
    hipLaunchKernelGGL(splitk_reduce, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, r);
    OTTER_CHECK_LAUNCH("otter_gemm (split-K reduce)");
    return OTTER_OK;
}

// The remainder rows of an in-place 16-bit residual product (x[m_main .. m_main + rem) += A W^T + bias), split-K, with the
// second pass fused into the row statistics of all m_main + rem rows of x (reduce_rows_stats).  A: the rem operand rows.
int otter_gemm64_splitk_stats_launch(int dtype, const void* A, long lda, const void* W, const float* bias, void* x, long ldx,
                                     int m_main, int rem, int N, int K, float* stats, float eps, float* workspace, int splits,
                                     hipStream_t s) {
    TailArgs a{(const char*)A, lda, (const char*)W, nullptr, nullptr, 0, nullptr, 0, rem, N, K, (rem + TB - 1) / TB, (N + TB - 1) / TB,
               nullptr, nullptr, splits, workspace};
    int rc = dtype == DT_BF16 ? launch_tail_ns<__bf16, ACT_NONE, true, false, 5>(a, s) : launch_tail_ns<_Float16, ACT_NONE, true, false, 5>(a, s);
    if (rc != OTTER_OK) return rc;
    ReduceStatsArgs r{workspace, splits, bias, x, ldx, stats, m_main, rem, N, eps};
    return dtype == DT_BF16 ? launch_reduce_rows_stats<__bf16>(r, s) : launch_reduce_rows_stats<_Float16>(r, s);
}

// Shapes the 64x64 kernel accepts: 8-column vector stores and 32-bit source offsets.
bool otter_gemm64_eligible(int dtype, long lda, int n_pad, int M, int N, int K, const void* C, long ldc) {
    const long es = dtype == DT_F32 ? 4 : 2;
    if (N % 8 != 0 || ((uintptr_t)C & 15) != 0) return false;
    if (ldc % (dtype == DT_F32 ? 4 : 8) != 0) return false;                       // 16-bit outputs are stored 8 at a time; fp32 ones 4 at a time
    if (n_pad < ((N + TB - 1) / TB) * TB) return false;
    return (long)M * lda * es < (1L << 32) && (long)n_pad * K * es < (1L << 32);
}

int otter_gemm64_launch(int dtype, const void* A, long lda, const void* W, const float* bias, const void* residual, long ldr,
                        void* C, long ldc, int M, int N, int K, int act, bool out_f32, bool res, const float* ln_stats,
                        const float* ln_colsum, hipStream_t s) {
    TailArgs a{(const char*)A, lda, (const char*)W, bias, residual, ldr, C, ldc, M, N, K, (M + TB - 1) / TB, (N + TB - 1) / TB, ln_stats, ln_colsum, 1, nullptr};
    if (dtype == DT_F32) return dispatch_tail_f32(a, act, res, s);
    return dtype == DT_BF16 ? dispatch_tail<__bf16>(a, act, out_f32, res, s) : dispatch_tail<_Float16>(a, act, out_f32, res, s);
}
