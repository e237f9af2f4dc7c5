// Row-wise LayerNorm kernels (memory-bound): one 64-lane wave per row, the row held in registers,
// 16-byte loads/stores, two-pass mean/variance in fp32 (matches torch.nn.LayerNorm numerics),
// input fp32 or a 16-bit type (the residual stream), output a 16-bit type (the next GEMM's A operand) or fp32.
//
// Three row sources share the kernel:
//   SRC_PLAIN   x[row]                                   (ln_1 / ln_2 / ln_post / ln_final / policy LNs)
//   SRC_VIT     row t of image i = (t == 0 ? class_embedding : patch_tok[i*P + t-1]) + pos[t]   (-> ln_pre)
//   SRC_TEXT    token_embedding[ids[b*ctx + t]] + pos[t]  (text tower input; no normalisation, fp32 out)
#include "common.hpp"
#include "otter_hip.h"

namespace {

enum { SRC_PLAIN = 0, SRC_VIT = 1, SRC_TEXT = 2 };
enum { DT_STATS = 3 };      // output code of otter_row_stats: y[row] = {mean, rstd}, nothing else is written

struct LnArgs {
    const void* x; long ldx;             // fp32, or (SRC_PLAIN only) bf16 / fp16
    const float* gamma; const float* beta;
    void* y; long ldy;
    int rows, width; float eps;
    // SRC_VIT
    const float* cls; const float* pos; int tokens;      // tokens = P + 1
    // SRC_TEXT
    const int* ids; int ctx, take, vocab;                 // row = b*take + t reads ids[b*ctx + t]
};

template <int NV, int SRC, int IN_DT, int OUT_DT, bool NORMALISE>
__global__ __launch_bounds__(256) void ln_rows(LnArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int nvec = p.width >> 2;

    static_assert(IN_DT == DT_F32 || SRC == SRC_PLAIN, "16-bit input: plain rows only");
    const float* src = nullptr;
    const float* add = nullptr;
    if (SRC == SRC_PLAIN) {
        if (IN_DT == DT_F32) src = (const float*)p.x + (long)row * p.ldx;
    } else if (SRC == SRC_VIT) {
        const int img = row / p.tokens, t = row % p.tokens;
        src = (t == 0) ? p.cls : (const float*)p.x + ((long)img * (p.tokens - 1) + (t - 1)) * p.ldx;
        add = p.pos + (long)t * p.width;
    } else {
        const int b = row / p.take, t = row % p.take;
        const int id = min(max(p.ids[(long)b * p.ctx + t], 0), p.vocab - 1);   // never index outside the table
        src = (const float*)p.x + (long)id * p.ldx;
        add = p.pos + (long)t * p.width;
    }

    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 64 + lane;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (idx < nvec) {
            if constexpr (IN_DT == DT_BF16) {
                const bf16x4 h = *(const bf16x4*)((const __bf16*)p.x + (long)row * p.ldx + idx * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = (float)h[e];
            } else if constexpr (IN_DT == DT_F16) {
                const f16x4 h = *(const f16x4*)((const _Float16*)p.x + (long)row * p.ldx + idx * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = (float)h[e];
            } else {
                v[i] = *(const f32x4*)(src + idx * 4);
            }
            if (SRC != SRC_PLAIN) v[i] += *(const f32x4*)(add + idx * 4);
            sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    float mean = 0.f, rstd = 1.f;
    if (NORMALISE) {
        mean = wave_sum(sum) / (float)p.width;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 64 + lane;
            if (idx < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { float d = v[i][e] - mean; sq += d * d; }
            }
        }
        rstd = rsqrtf(wave_sum(sq) / (float)p.width + p.eps);
    }
    if constexpr (OUT_DT == DT_STATS) {
        if (lane == 0) *(f32x2*)((float*)p.y + 2 * (long)row) = (f32x2){mean, rstd};
        return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 64 + lane;
        if (idx >= nvec) continue;
        f32x4 o = v[i];
        if (NORMALISE) {
            o = (v[i] - mean) * rstd;
            if (p.gamma) o = o * *(const f32x4*)(p.gamma + idx * 4) + *(const f32x4*)(p.beta + idx * 4);
        }
        if constexpr (OUT_DT == DT_F32) {
            *(f32x4*)((float*)p.y + (long)row * p.ldy + idx * 4) = o;
        } else if constexpr (OUT_DT == DT_BF16) {
            bf16x4 ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[e] = Half16<__bf16>::from(o[e]);
            *(bf16x4*)((__bf16*)p.y + (long)row * p.ldy + idx * 4) = ob;
        } else {
            f16x4 ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[e] = Half16<_Float16>::from(o[e]);
            *(f16x4*)((_Float16*)p.y + (long)row * p.ldy + idx * 4) = ob;
        }
    }
}

// Row statistics of a 16-bit residual stream (width <= 1024, a multiple of 8): the only work is one read of x, taken 16
// bytes per lane and load (ln_rows reads 8).  R rows per wave, all their loads in flight before the first reduction;
// measured on 32896 x 1024 fp16: R = 1: 11.7 us (5.7 TB/s), 2: 12.2, 3: 12.9, 4: 13.7, 8: 17.5; ln_rows: 15.9.
// Two-pass mean / variance on the registers, as ln_rows.
template <typename T, int R>
__global__ __launch_bounds__(256) void row_stats16(const T* x, long ldx, float* stats, int rows, int width, float eps) {
    typedef typename Half16<T>::v8 v8;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int nvec = width >> 3;
    v8 v[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const T* src = x + (long)min(row0 + r, rows - 1) * ldx;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = i * 64 + lane;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[r][i][e] = Half16<T>::from(0.f);
            if (idx < nvec) v[r][i] = *(const v8*)(src + idx * 8);
        }
    }
    const float inv_w = 1.0f / (float)width;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const f32x2 st = row_mean_rstd_16<T>(v[r], lane, nvec, inv_w, eps);
        if (lane == 0 && row0 + r < rows) *(f32x2*)(stats + 2 * (long)(row0 + r)) = st;
    }
}

template <int SRC, int IN_DT, int OUT_DT, bool NORMALISE>
int launch_ln(const LnArgs& a, hipStream_t s, const char* name) {
    const int grid = (a.rows + 3) / 4;
    if (a.width <= 1024) hipLaunchKernelGGL((ln_rows<4, SRC, IN_DT, OUT_DT, NORMALISE>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ln_rows<16, SRC, IN_DT, OUT_DT, NORMALISE>), dim3(grid), dim3(256), 0, s, a);
    OTTER_CHECK_LAUNCH(name);
    return OTTER_OK;
}

template <int SRC, int IN_DT, bool NORMALISE>
int launch_ln_out(const LnArgs& a, int out_dtype, hipStream_t s, const char* name) {
    if (out_dtype == DT_F32) return launch_ln<SRC, IN_DT, DT_F32, NORMALISE>(a, s, name);
    if (out_dtype == DT_BF16) return launch_ln<SRC, IN_DT, DT_BF16, NORMALISE>(a, s, name);
    if (out_dtype == DT_F16) return launch_ln<SRC, IN_DT, DT_F16, NORMALISE>(a, s, name);
    otter_set_error("%s: unknown out_dtype %d", name, out_dtype);
    return OTTER_ERR_INVALID_ARGUMENT;
}

int check_width(int width, const char* name) {
    OTTER_REQUIRE(width > 0 && width % 4 == 0 && width <= 4096, OTTER_ERR_UNSUPPORTED_SHAPE,
                  "%s: width=%d must be a multiple of 4 and <= 4096", name, width);
    return OTTER_OK;
}

}  // namespace

extern "C" int otter_layernorm(const void* x, int in_dtype, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                               int rows, int width, float eps, int out_dtype, void* stream) {
    OTTER_REQUIRE(x && y && ((gamma == nullptr) == (beta == nullptr)), OTTER_ERR_INVALID_ARGUMENT,
                  "otter_layernorm: null operand (gamma and beta must both be given or both be NULL)");
    if (rows == 0) return OTTER_OK;
    OTTER_REQUIRE(rows > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_layernorm: rows=%d", rows);
    if (int rc = check_width(width, "otter_layernorm")) return rc;
    OTTER_REQUIRE(ldx >= width && ldx % 4 == 0 && ldy >= width && ldy % 4 == 0, OTTER_ERR_ALIGNMENT,
                  "otter_layernorm: leading dimensions must be >= width and multiples of 4");
    OTTER_REQUIRE(aligned16(x) && (!gamma || (aligned16(gamma) && aligned16(beta))) && (reinterpret_cast<uintptr_t>(y) & 7u) == 0,
                  OTTER_ERR_ALIGNMENT, "otter_layernorm: operands must be 16-byte aligned");
    LnArgs a{};
    a.x = x; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.y = y; a.ldy = ldy; a.rows = rows; a.width = width; a.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == DT_F32) return launch_ln_out<SRC_PLAIN, DT_F32, true>(a, out_dtype, s, "otter_layernorm");
    if (in_dtype == DT_BF16) return launch_ln_out<SRC_PLAIN, DT_BF16, true>(a, out_dtype, s, "otter_layernorm");
    if (in_dtype == DT_F16) return launch_ln_out<SRC_PLAIN, DT_F16, true>(a, out_dtype, s, "otter_layernorm");
    otter_set_error("otter_layernorm: unknown in_dtype %d", in_dtype);
    return OTTER_ERR_INVALID_ARGUMENT;
}

extern "C" int otter_vit_embed_lnpre(const float* patch_tok, long ld_tok, const float* class_embedding, const float* pos_embedding,
                                     const float* gamma, const float* beta, void* x, int out_dtype, long ldx, int n_img,
                                     int n_patches, int width, float eps, void* stream) {
    OTTER_REQUIRE(patch_tok && class_embedding && pos_embedding && gamma && beta && x, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_vit_embed_lnpre: null operand");
    if (n_img == 0) return OTTER_OK;
    OTTER_REQUIRE(n_img > 0 && n_patches > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_vit_embed_lnpre: n_img=%d n_patches=%d", n_img, n_patches);
    if (int rc = check_width(width, "otter_vit_embed_lnpre")) return rc;
    OTTER_REQUIRE(ld_tok >= width && ld_tok % 4 == 0 && ldx >= width && ldx % 4 == 0, OTTER_ERR_ALIGNMENT,
                  "otter_vit_embed_lnpre: leading dimensions must be >= width and multiples of 4");
    OTTER_REQUIRE(aligned16(patch_tok) && aligned16(class_embedding) && aligned16(pos_embedding) && aligned16(x) &&
                  aligned16(gamma) && aligned16(beta), OTTER_ERR_ALIGNMENT, "otter_vit_embed_lnpre: operands must be 16-byte aligned");
    LnArgs a{};
    a.x = patch_tok; a.ldx = ld_tok; a.gamma = gamma; a.beta = beta; a.y = x; a.ldy = ldx;
    a.rows = n_img * (n_patches + 1); a.width = width; a.eps = eps;
    a.cls = class_embedding; a.pos = pos_embedding; a.tokens = n_patches + 1;
    return launch_ln_out<SRC_VIT, DT_F32, true>(a, out_dtype, (hipStream_t)stream, "otter_vit_embed_lnpre");
}

extern "C" int otter_text_embed(const int* ids, int ctx, int take, const float* token_embedding, int vocab,
                                const float* pos_embedding, void* x, int out_dtype, long ldx, int batch, int width, void* stream) {
    OTTER_REQUIRE(ids && token_embedding && pos_embedding && x, OTTER_ERR_INVALID_ARGUMENT, "otter_text_embed: null operand");
    if (batch == 0) return OTTER_OK;
    OTTER_REQUIRE(batch > 0 && take > 0 && take <= ctx && vocab > 0, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_text_embed: batch=%d take=%d ctx=%d vocab=%d", batch, take, ctx, vocab);
    if (int rc = check_width(width, "otter_text_embed")) return rc;
    OTTER_REQUIRE(ldx >= width && ldx % 4 == 0, OTTER_ERR_ALIGNMENT, "otter_text_embed: ldx must be >= width and a multiple of 4");
    OTTER_REQUIRE(aligned16(token_embedding) && aligned16(pos_embedding) && aligned16(x), OTTER_ERR_ALIGNMENT,
                  "otter_text_embed: operands must be 16-byte aligned");
    LnArgs a{};
    a.x = token_embedding; a.ldx = width; a.y = x; a.ldy = ldx; a.rows = batch * take; a.width = width;
    a.pos = pos_embedding; a.ids = ids; a.ctx = ctx; a.take = take; a.vocab = vocab;
    return launch_ln_out<SRC_TEXT, DT_F32, false>(a, out_dtype, (hipStream_t)stream, "otter_text_embed");
}

extern "C" int otter_row_stats(const void* x, int in_dtype, long ldx, float* stats, int rows, int width, float eps, void* stream) {
    OTTER_REQUIRE(x && stats, OTTER_ERR_INVALID_ARGUMENT, "otter_row_stats: null operand");
    if (rows == 0) return OTTER_OK;
    OTTER_REQUIRE(rows > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_row_stats: rows=%d", rows);
    if (int rc = check_width(width, "otter_row_stats")) return rc;
    OTTER_REQUIRE(ldx >= width && ldx % 4 == 0 && aligned16(x) && (reinterpret_cast<uintptr_t>(stats) & 7u) == 0, OTTER_ERR_ALIGNMENT,
                  "otter_row_stats: x must be 16-byte aligned with ldx a multiple of 4, stats 8-byte aligned");
    LnArgs a{};
    a.x = x; a.ldx = ldx; a.y = stats; a.ldy = 2; a.rows = rows; a.width = width; a.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    if ((in_dtype == DT_BF16 || in_dtype == DT_F16) && width % 8 == 0 && width <= 1024 && ldx % 8 == 0) {
        constexpr int R = 1;
        const int grid = (rows + 4 * R - 1) / (4 * R);
        if (in_dtype == DT_BF16) hipLaunchKernelGGL((row_stats16<__bf16, R>), dim3(grid), dim3(256), 0, s, (const __bf16*)x, ldx, stats, rows, width, eps);
        else hipLaunchKernelGGL((row_stats16<_Float16, R>), dim3(grid), dim3(256), 0, s, (const _Float16*)x, ldx, stats, rows, width, eps);
        OTTER_CHECK_LAUNCH("otter_row_stats");
        return OTTER_OK;
    }
    if (in_dtype == DT_F32) return launch_ln<SRC_PLAIN, DT_F32, DT_STATS, true>(a, s, "otter_row_stats");
    if (in_dtype == DT_BF16) return launch_ln<SRC_PLAIN, DT_BF16, DT_STATS, true>(a, s, "otter_row_stats");
    if (in_dtype == DT_F16) return launch_ln<SRC_PLAIN, DT_F16, DT_STATS, true>(a, s, "otter_row_stats");
    otter_set_error("otter_row_stats: unknown in_dtype %d", in_dtype);
    return OTTER_ERR_INVALID_ARGUMENT;
}
