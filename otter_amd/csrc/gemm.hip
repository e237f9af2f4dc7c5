// MFMA GEMM with fused epilogues:  C[M,N] = epi(A[M,K] * W[N,K]^T + bias [+ residual]).
//
// Operand types: bf16 or fp16 (v_mfma_f32_16x16x32_*; the CLIP towers) and fp32 (v_mfma_f32_16x16x4_f32, an exact
// fp32 fma chain; the small policy / pooling / head layers whose outputs must match an fp32 reference closely).
// Accumulation is fp32 in every case.
//
// Both operands are K-contiguous (activations row-major, weights in torch's [out,in] layout), so the
// same LDS image and fragment read serve both.  Structure (one workgroup = 4 waves = 128x128 output tile):
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), two LDS stages so
//     the load of K-step t+1 overlaps the MFMAs of step t; a K-step is 128 bytes of K per row (64 halves / 32 floats);
//   * the LDS image is lane-linear as LDS-DMA requires; bank conflicts are removed by an XOR swizzle of the
//     16-byte chunk index applied to the per-lane SOURCE address and again on the ds_read_b128 address;
//   * 4x4 accumulator tiles of 16x16 per wave (64x64);
//   * for fp32 operands a lane's 16-byte chunk holds 4 consecutive k; element e of every lane feeds MFMA step e,
//     which permutes k identically for both operands;
//   * the weight rows are the MFMA "A" operand and the activation rows the "B" operand, so an accumulator
//     register quad holds 4 consecutive output columns of one output row -> 8/16-byte epilogue stores;
//   * workgroup ids are remapped per XCD and walked in groups of 8 row-tiles so concurrently resident
//     tiles share A/W panels in the XCD's L2.
#include "common.hpp"
#include "otter_hip.h"

#include <map>
#include <mutex>

// 256x256-tile kernel for the big 16-bit shapes (gemm256.hip)
bool otter_gemm256_eligible(int dtype, long lda, int n_pad, int M, int N, int K, const void* C, long ldc);
int otter_gemm256_launch(int dtype, const void* A, long lda, const void* W, const float* bias, const void* residual, long ldr,
                         void* C, long ldc, int M, int N, int K, int act, bool out_f32, bool res, const float* ln_stats,
                         const float* ln_colsum, hipStream_t s);

// 64x64-tile kernel for remainder rows and small problems (gemm_tail.hip)
bool otter_gemm64_eligible(int dtype, long lda, int n_pad, int M, int N, int K, const void* C, long ldc);
int otter_gemm64_launch(int dtype, const void* A, long lda, const void* W, const float* bias, const void* residual, long ldr,
                        void* C, long ldc, int M, int N, int K, int act, bool out_f32, bool res, const float* ln_stats,
                        const float* ln_colsum, hipStream_t s);

int otter_gemm64_splitk_stats_launch(int dtype, const void* A, long lda, const void* W, const float* bias, void* x, long ldx,
                                     int m_main, int rem, int N, int K, float* stats, float eps, float* workspace, int splits,
                                     hipStream_t s);
int otter_gemm64_splitk_launch(int dtype, const void* A, long lda, const void* W, const float* bias, const void* residual, long ldr,
                               void* C, long ldc, int M, int N, int K, int act, bool out_f32, bool res, const float* ln_stats,
                               const float* ln_colsum, float* workspace, int splits, hipStream_t s);

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 128;                           // bytes of K per tile row and K-step
constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;       // 32 KiB
constexpr int GROUP_M = 8;

struct GemmArgs {
    const char* A; long lda;    // lda, ldw in elements
    const char* W;              // [n_pad][K]
    const float* bias;          // [N] or null
    const void* residual; long ldr;     // in the OUTPUT's type
    void* C; long ldc;
    int M, N, K, tiles_m, tiles_n;
    const float* ln_stats;      // [M][2] (mean, rstd) or null: LayerNorm folded into the epilogue (see otter_gemm_ln)
    const float* ln_colsum;     // [N]
};

enum { ACT_NONE = 0, ACT_QUICKGELU = 1, ACT_GELU = 2, ACT_RELU = 3 };

template <int ACT> __device__ __forceinline__ float apply_act(float v) {
    if (ACT == ACT_QUICKGELU) return quick_gelu(v);
    if (ACT == ACT_GELU) return gelu_erf(v);
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

template <typename T> struct Frag { typedef typename Half16<T>::v8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ f32x4 mma(const typename Frag<T>::type& w, const typename Frag<T>::type& a, f32x4 c) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], a[e], c, 0, 0, 0);
        return c;
    } else {
        return Half16<T>::mfma32(w, a, c);
    }
}

template <typename T, int ACT, bool OUT_F32, bool RES>
__global__ __launch_bounds__(256, 2) void gemm_128x128(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Frag<T>::type frag_t;
    constexpr int ES = sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;

    const int nwg = p.tiles_m * p.tiles_n;
    const int pid = xcd_remap(blockIdx.x, nwg);
    const int width = GROUP_M * p.tiles_n;
    const int first_m = (pid / width) * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (pid % width) % gsz;
    const int tn = (pid % width) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging source addresses (4 LDS-DMA instructions per operand per wave and K-step) ----
    // instruction i of wave w fills LDS rows (4w+i)*8 .. +7; lane l -> row +(l>>3), physical chunk l&7,
    // which holds logical chunk (l&7) ^ ((row>>1)&7).
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int am = min(m0 + r, p.M - 1);                 // clamp: rows past M are never stored
        a_src[i] = p.A + ((long)am * p.lda) * ES + c * 16;
        w_src[i] = p.W + ((long)(n0 + r) * p.K) * ES + c * 16;   // weights are padded to a tile multiple
    }

    // ---- fragment read offsets ----
    const int swz = (lane >> 1) & 7;                         // ((row>>1)&7) with row = 16*t + (lane&15)
    const int row_off = (lane & 15) * ROW_BYTES;
    int frag_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) frag_off[ks] = row_off + ((((ks * 4) + (lane >> 4)) ^ swz) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, long kbyte) {
        char* sA = smem + buf * STAGE_BYTES;
        char* sW = sA + BM * ROW_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a_src[i] + kbyte), LDS_PTR(sA + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(w_src[i] + kbyte), LDS_PTR(sW + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };

    const int nk = (p.K * ES) / ROW_BYTES;
    stage(0, 0);
    __syncthreads();                                          // drains the LDS-DMA (vmcnt(0)) + barrier

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) stage(cur ^ 1, (long)(t + 1) * ROW_BYTES);
        const char* sA = smem + cur * STAGE_BYTES + wm * 64 * ROW_BYTES;
        const char* sW = smem + cur * STAGE_BYTES + BM * ROW_BYTES + wn * 64 * ROW_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t wf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *(const frag_t*)(sW + i * 16 * ROW_BYTES + frag_off[ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = *(const frag_t*)(sA + j * 16 * ROW_BYTES + frag_off[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma<T>(wf[i], af[j], acc[i][j]);
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = n][col = m]; lane holds m = lane&15, n = 4*(lane>>4) + 0..3 ----
    const int nq = (lane >> 4) * 4, mc = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + nq;
        if (n >= p.N) continue;
        f32x4 b = {0.f, 0.f, 0.f, 0.f}, cs = b;
        if (p.bias) b = *(const f32x4*)(p.bias + n);
        if (p.ln_stats) cs = *(const f32x4*)(p.ln_colsum + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + mc;
            if (m >= p.M) continue;
            f32x4 v = acc[i][j];
            if (p.ln_stats) {
                const f32x2 st = *(const f32x2*)(p.ln_stats + 2 * (long)m);
                v = (v - st[0] * cs) * st[1];
            }
            v += b;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act<ACT>(v[e]);
            if (RES) {
                if constexpr (OUT_F32 || sizeof(T) == 4) {
                    v += *(const f32x4*)((const float*)p.residual + (long)m * p.ldr + n);
                } else {
                    const typename Half16<T>::v4 r = *(const typename Half16<T>::v4*)((const T*)p.residual + (long)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
                }
            }
            if constexpr (OUT_F32 || sizeof(T) == 4) {
                *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = v;
            } else {
                typename Half16<T>::v4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = Half16<T>::from(v[e]);
                *(typename Half16<T>::v4*)((T*)p.C + (long)m * p.ldc + n) = o;
            }
        }
    }
}

template <typename T, int ACT, bool OUT_F32, bool RES>
int launch(const GemmArgs& a, hipStream_t s) {
    auto kern = gemm_128x128<T, ACT, OUT_F32, RES>;
    if (int rc = otter_reserve_lds((const void*)kern, 2 * STAGE_BYTES, "otter_gemm (128x128)")) return rc;
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), 2 * STAGE_BYTES, s, a);
    OTTER_CHECK_LAUNCH("otter_gemm");
    return OTTER_OK;
}

template <typename T, int ACT>
int pick_out(const GemmArgs& a, bool out_f32, hipStream_t s) {
    if constexpr (sizeof(T) == 4) {
        return launch<T, ACT, true, false>(a, s);
    } else {
        return out_f32 ? launch<T, ACT, true, false>(a, s) : launch<T, ACT, false, false>(a, s);
    }
}

template <typename T>
int dispatch(const GemmArgs& a, int act, bool out_f32, bool res, hipStream_t s) {
    if constexpr (sizeof(T) == 4) {
        if (res) return launch<T, ACT_NONE, true, true>(a, s);
    } else {
        if (res) return out_f32 ? launch<T, ACT_NONE, true, true>(a, s) : launch<T, ACT_NONE, false, true>(a, s);
    }
    switch (act) {
        case OTTER_EPI_ACT_NONE: return pick_out<T, ACT_NONE>(a, out_f32, s);
        case OTTER_EPI_ACT_QUICKGELU: return pick_out<T, ACT_QUICKGELU>(a, out_f32, s);
        case OTTER_EPI_ACT_GELU: return pick_out<T, ACT_GELU>(a, out_f32, s);
        case OTTER_EPI_ACT_RELU: return pick_out<T, ACT_RELU>(a, out_f32, s);
    }
    otter_set_error("otter_gemm: unknown activation %d", act);
    return OTTER_ERR_INVALID_ARGUMENT;
}

}  // namespace

// Split-K factor for a 16-bit problem on the 64x64 kernel: few tiles (each would pull its whole K panel through one CU)
// and a K loop long enough to cut; 1 = do not split.  About two workgroups per CU at most, and only when the tiles
// alone cover at most half the CUs: from there on the launch is bound by the bytes all tiles pull through the L2, which
// splitting does not reduce, and the second launch is pure cost (measured on 514 rows x 1024 columns, 144 tiles: 14.9 us +
// 4.8 us split in two against 12 us in one pass).
static int pick_splits(int dtype, int M, int N, int K, const float* workspace, long workspace_floats) {
    if (!workspace || N % 8 != 0) return 1;
    const int nk = K / (dtype == DT_F32 ? 32 : 64), tiles = ((M + 63) / 64) * ((N + 63) / 64);
    // fp32 (policy transformer, heads): only the few-row launches of a rollout step with one or a few episodes, whose K loop
    // is a chain of 24+ dependent K-steps on a handful of CUs; with more tiles the fp32 MFMA rate is the bound
    // 16-bit: from ~96 tiles on one pass is as fast as two launches (the 128 class-token rows of a 64-observation step against
    // N = 3072 / 4096: the step is 0.25 % faster unsplit; against N = 1024, 32 tiles, it is slower unsplit)
    if (tiles > (dtype == DT_F32 ? 48 : 64)) return 1;
    for (int sp = 8; sp > 1; sp >>= 1)
        if (nk % sp == 0 && nk / sp >= 2 && tiles * sp <= 512 && (long)sp * M * N <= workspace_floats) return sp;
    return 1;
}

static int gemm_impl(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias,
                     const void* residual, long ldr, void* C, long ldc, int M, int N, int K,
                     int epilogue, const float* ln_stats, const float* ln_colsum, float* workspace, long workspace_floats, void* stream) {
    OTTER_REQUIRE(dtype == DT_BF16 || dtype == DT_F16 || dtype == DT_F32, OTTER_ERR_INVALID_ARGUMENT, "otter_gemm: unknown dtype %d", dtype);
    const int es = dtype == DT_F32 ? 4 : 2;
    const int kq = ROW_BYTES / es;
    OTTER_REQUIRE(A && W && C, OTTER_ERR_INVALID_ARGUMENT, "otter_gemm: null operand");
    OTTER_REQUIRE(M > 0 && N > 0 && K > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_gemm: empty problem M=%d N=%d K=%d", M, N, K);
    OTTER_REQUIRE(K % kq == 0, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_gemm: K=%d must be a multiple of %d (pad the operands)", K, kq);
    OTTER_REQUIRE(N % 4 == 0, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_gemm: N=%d must be a multiple of 4", N);
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    OTTER_REQUIRE(n_pad >= tiles_n * BN, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_gemm: weight has %d rows, needs >= %d (pad N to a multiple of %d)", n_pad, tiles_n * BN, BN);
    OTTER_REQUIRE(lda >= K && (lda * es) % 16 == 0, OTTER_ERR_ALIGNMENT, "otter_gemm: lda=%ld must be >= K and give 16-byte aligned rows", lda);
    const bool out_f32 = (epilogue & OTTER_EPI_OUT_F32) != 0 || dtype == DT_F32, res = (epilogue & OTTER_EPI_RESIDUAL) != 0;
    const int act = epilogue & OTTER_EPI_ACT_MASK;
    OTTER_REQUIRE(ldc >= N && ldc % 4 == 0, OTTER_ERR_ALIGNMENT, "otter_gemm: ldc=%ld must be >= N and a multiple of 4", ldc);
    OTTER_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C) && (!bias || aligned16(bias)), OTTER_ERR_ALIGNMENT,
                  "otter_gemm: operands must be 16-byte aligned");
    OTTER_REQUIRE((long)tiles_m * tiles_n < (1L << 31), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_gemm: too many tiles");
    if (res) {
        OTTER_REQUIRE(residual && aligned16(residual) && ldr >= N && ldr % (out_f32 ? 4 : 8) == 0, OTTER_ERR_INVALID_ARGUMENT,
                      "otter_gemm: residual epilogue needs a 16-byte aligned residual of the output's type with ldr >= N");
        OTTER_REQUIRE(act == OTTER_EPI_ACT_NONE, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_gemm: residual epilogue takes no activation");
    }
    hipStream_t s = (hipStream_t)stream;
    // (the 256x256 kernel has the folded-LayerNorm epilogue for 16-bit outputs with no / QuickGELU / GELU activation only)
    // (... and it fetches the row statistics 16 bytes at a time)
    const bool ln256 = !ln_stats || (!out_f32 && act != OTTER_EPI_ACT_RELU && aligned16(ln_stats));
    if (!(epilogue & OTTER_EPI_TILE_128) && ln256 && otter_gemm256_eligible(dtype, lda, n_pad, M, N, K, C, ldc)) {
        // Rows are independent: whole 256-row tiles go to the 256x256 kernel, the remaining (< 256) rows to the
        // 64x64 kernel.
        const int m_main = (M / 256) * 256;
        int rc = otter_gemm256_launch(dtype, A, lda, W, bias, residual, ldr, C, ldc, m_main, N, K, act, out_f32, res, ln_stats, ln_colsum, s);
        if (rc != OTTER_OK || m_main == M) return rc;
        const long ec = (out_f32 ? 4 : es);
        const int rem = M - m_main;
        const int splits = pick_splits(dtype, rem, N, K, workspace, workspace_floats);
        if (splits > 1)
            return otter_gemm64_splitk_launch(dtype, (const char*)A + (long)m_main * lda * es, lda, W, bias,
                                              res ? (const char*)residual + (long)m_main * ldr * ec : nullptr, ldr,
                                              (char*)C + (long)m_main * ldc * ec, ldc, rem, N, K, act, out_f32, res,
                                              ln_stats ? ln_stats + 2 * (long)m_main : nullptr, ln_colsum, workspace, splits, s);
        return otter_gemm64_launch(dtype, (const char*)A + (long)m_main * lda * es, lda, W, bias,
                                      res ? (const char*)residual + (long)m_main * ldr * ec : nullptr, ldr, (char*)C + (long)m_main * ldc * ec, ldc,
                                      M - m_main, N, K, act, out_f32, res, ln_stats ? ln_stats + 2 * (long)m_main : nullptr, ln_colsum, s);
    }
    // too few 128x128 tiles to give every CU one: the 64x64 kernel spreads the problem over 4x more workgroups
    if (!(epilogue & OTTER_EPI_TILE_128) && (long)tiles_m * tiles_n < 256 && otter_gemm64_eligible(dtype, lda, n_pad, M, N, K, C, ldc)) {
        const int splits = pick_splits(dtype, M, N, K, workspace, workspace_floats);
        if (splits > 1)
            return otter_gemm64_splitk_launch(dtype, A, lda, W, bias, residual, ldr, C, ldc, M, N, K, act, out_f32, res, ln_stats, ln_colsum,
                                              workspace, splits, s);
        return otter_gemm64_launch(dtype, A, lda, W, bias, residual, ldr, C, ldc, M, N, K, act, out_f32, res, ln_stats, ln_colsum, s);
    }
    const int tm = (M + BM - 1) / BM;
    GemmArgs a{(const char*)A, lda, (const char*)W, bias, residual, ldr, C, ldc, M, N, K, tm, tiles_n, ln_stats, ln_colsum};
    if (dtype == DT_BF16) return dispatch<__bf16>(a, act, out_f32, res, s);
    if (dtype == DT_F16) return dispatch<_Float16>(a, act, out_f32, res, s);
    return dispatch<float>(a, act, true, res, s);
}

extern "C" int otter_gemm(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias,
                          const void* residual, long ldr, void* C, long ldc, int M, int N, int K,
                          int epilogue, void* stream) {
    return gemm_impl(dtype, A, lda, W, n_pad, bias, residual, ldr, C, ldc, M, N, K, epilogue, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int otter_gemm_ln(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias,
                             const float* row_stats, const float* col_sum, void* C, long ldc, int M, int N, int K,
                             int epilogue, void* stream) {
    OTTER_REQUIRE(row_stats && col_sum && aligned16(col_sum) && (reinterpret_cast<uintptr_t>(row_stats) & 7u) == 0, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_gemm_ln: row_stats (8-byte aligned) and col_sum (16-byte aligned) are required");
    OTTER_REQUIRE(!(epilogue & OTTER_EPI_RESIDUAL), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_gemm_ln: no residual epilogue");
    return gemm_impl(dtype, A, lda, W, n_pad, bias, nullptr, 0, C, ldc, M, N, K, epilogue, row_stats, col_sum, nullptr, 0, stream);
}

extern "C" long otter_gemm_workspace_floats(int N) { return 8L * 255 * N; }

extern "C" int otter_gemm_ws(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias, const void* residual,
                             long ldr, const float* row_stats, const float* col_sum, void* C, long ldc, int M, int N, int K,
                             int epilogue, float* workspace, long workspace_floats, void* stream) {
    OTTER_REQUIRE((row_stats == nullptr) == (col_sum == nullptr), OTTER_ERR_INVALID_ARGUMENT, "otter_gemm_ws: row_stats and col_sum go together");
    OTTER_REQUIRE(!row_stats || !(epilogue & OTTER_EPI_RESIDUAL), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_gemm_ws: folded LayerNorm takes no residual");
    OTTER_REQUIRE(!workspace || (aligned16(workspace) && workspace_floats > 0), OTTER_ERR_ALIGNMENT, "otter_gemm_ws: workspace must be 16-byte aligned");
    return gemm_impl(dtype, A, lda, W, n_pad, bias, residual, ldr, C, ldc, M, N, K, epilogue, row_stats, col_sum, workspace, workspace_floats, stream);
}

// x += A W^T + bias in place on a 16-bit residual stream, and (mean, rstd) of every updated row: otter_gemm_ws with the
// residual epilogue followed by otter_row_stats, bit for bit -- but when the < 256 remainder rows go split-K, their second
// pass shares one launch with the statistics of all rows (one launch less per residual product).
extern "C" int otter_gemm_residual_stats(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias, void* x,
                                         long ldx, int M, int N, int K, float* stats, float eps, float* workspace,
                                         long workspace_floats, void* stream) {
    OTTER_REQUIRE(dtype == DT_BF16 || dtype == DT_F16, OTTER_ERR_INVALID_ARGUMENT, "otter_gemm_residual_stats: dtype %d (a 16-bit stream)", dtype);
    OTTER_REQUIRE(x && stats && (reinterpret_cast<uintptr_t>(stats) & 15u) == 0, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_gemm_residual_stats: x and stats (16-byte aligned) are required");
    OTTER_REQUIRE(!workspace || (aligned16(workspace) && workspace_floats > 0), OTTER_ERR_ALIGNMENT, "otter_gemm_residual_stats: workspace must be 16-byte aligned");
    OTTER_REQUIRE(N > 0 && N <= 1024 && N % 8 == 0, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_gemm_residual_stats: N=%d (a multiple of 8, at most 1024)", N);
    hipStream_t s = (hipStream_t)stream;
    const int m_main = (M / 256) * 256, rem = M - m_main;
    const int splits = (m_main > 0 && rem > 0 && K > 0 && lda >= K && otter_gemm256_eligible(dtype, lda, n_pad, M, N, K, x, ldx))
                           ? pick_splits(dtype, rem, N, K, workspace, workspace_floats) : 1;
    if (splits > 1 && ldx % 8 == 0 && aligned16(x) && aligned16(A) && aligned16(W) && (!bias || aligned16(bias)) &&
        n_pad >= ((N + BN - 1) / BN) * BN && (lda * 2) % 16 == 0) {
        int rc = otter_gemm256_launch(dtype, A, lda, W, bias, x, ldx, x, ldx, m_main, N, K, OTTER_EPI_ACT_NONE, false, true, nullptr, nullptr, s);
        if (rc != OTTER_OK) return rc;
        return otter_gemm64_splitk_stats_launch(dtype, (const char*)A + (long)m_main * lda * 2, lda, W, bias, x, ldx, m_main, rem, N, K,
                                                stats, eps, workspace, splits, s);
    }
    int rc = gemm_impl(dtype, A, lda, W, n_pad, bias, x, ldx, x, ldx, M, N, K, OTTER_EPI_RESIDUAL, nullptr, nullptr, workspace, workspace_floats, stream);
    if (rc != OTTER_OK) return rc;
    return otter_row_stats(x, dtype, ldx, stats, M, N, eps, stream);
}

// ---- allocation-free variant of otter_gemm_ws: the split-K scratch is a static device array ---------------------
// One copy of the array lives on every GPU the library is loaded on, until the process ends.  Host code must not pass
// the array's NAME to a kernel (on the host that is a stand-in object, not a device address): the address of the
// current device's copy is looked up once per device with hipGetSymbolAddress and only that pointer is used.
constexpr int STATIC_WS_MAX_N = 4096;
constexpr long STATIC_WS_FLOATS = 8L * 255 * STATIC_WS_MAX_N;      // == otter_gemm_workspace_floats(4096): 33.4 MB
static __device__ __align__(256) float scratch[STATIC_WS_FLOATS];

// -> the current device's copy of `scratch` (looked up on the device's first call, under a lock), or nullptr with the error
// text set.  `stream`'s device must be the current one: the pointer is that device's, and so is the launch.
static float* static_scratch_of_current_device(hipStream_t stream, const char* who) {
    static std::mutex mu;
    static std::map<int, float*> per_device;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        otter_set_error("%s: no current HIP device", who);
        return nullptr;
    }
    if (stream) {
        int sdev = -1;
        if (hipStreamGetDevice(stream, &sdev) != hipSuccess || sdev != dev) {
            otter_set_error("%s: the stream belongs to device %d, the current device is %d (make the stream's device current)", who, sdev, dev);
            return nullptr;
        }
    }
    std::lock_guard<std::mutex> lock(mu);
    auto it = per_device.find(dev);
    if (it != per_device.end()) return it->second;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        otter_set_error("%s: first use on device %d while the stream is being captured: call otter_gemm_static_ws_init() on this device "
                        "before the capture (the symbol lookup may load the code object, which a capture does not allow)", who, dev);
        return nullptr;
    }
    float* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
        otter_set_error("%s: hipGetSymbolAddress(scratch) failed on device %d: %s", who, dev, hipGetErrorString(e));
        return nullptr;
    }
    per_device[dev] = ptr;
    return ptr;
}

extern "C" int otter_gemm_static_ws_init(void) {
    return static_scratch_of_current_device(nullptr, "otter_gemm_static_ws_init") ? OTTER_OK : OTTER_ERR_LAUNCH;
}

extern "C" long otter_gemm_static_ws_max_n(void) { return STATIC_WS_MAX_N; }

extern "C" int otter_gemm_static_ws(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias, const void* residual,
                                    long ldr, const float* row_stats, const float* col_sum, void* C, long ldc, int M, int N, int K,
                                    int epilogue, void* stream) {
    OTTER_REQUIRE((row_stats == nullptr) == (col_sum == nullptr), OTTER_ERR_INVALID_ARGUMENT, "otter_gemm_static_ws: row_stats and col_sum go together");
    OTTER_REQUIRE(!row_stats || !(epilogue & OTTER_EPI_RESIDUAL), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_gemm_static_ws: folded LayerNorm takes no residual");
    OTTER_REQUIRE(N <= STATIC_WS_MAX_N, OTTER_ERR_UNSUPPORTED_SHAPE,
                  "otter_gemm_static_ws: N=%d exceeds the static scratch (N <= %d); use otter_gemm_ws with a caller-provided workspace", N,
                  STATIC_WS_MAX_N);
    float* ws = static_scratch_of_current_device((hipStream_t)stream, "otter_gemm_static_ws");
    if (!ws) return OTTER_ERR_LAUNCH;
    // (the split-K pass writes every partial product it later reads, so the scratch needs no clearing between calls)
    return gemm_impl(dtype, A, lda, W, n_pad, bias, residual, ldr, C, ldc, M, N, K, epilogue, row_stats, col_sum, ws, STATIC_WS_FLOATS, stream);
}
