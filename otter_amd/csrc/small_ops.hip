// Small memory-bound helpers around the big kernels: patch extraction (im2col with the CLIP pixel
// normalisation fused), row L2-normalisation, fp32 small linear layers (proprioception MLP), casts,
// row broadcast/gather, and the GPU-resident rollout window of policy inputs.
#include "common.hpp"
#include "otter_hip.h"

namespace {

// ---------------------------------------------------------------- im2col (+ /255, (x-mean)/std for uint8 input)
// out[(img*g*g + py*g + px)][c*ps*ps + ky*ps + kx] = norm(img[c][py*ps+ky][px*ps+kx]); columns >= 3*ps*ps are zero.
template <bool U8, typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const void* images, T* out, int res, int ps, int g, int kpad,
                                                     float m0, float m1, float m2, float s0, float s1, float s2) {
    const int patch = blockIdx.x;                      // img*g*g + py*g + px
    const int img = patch / (g * g), py = (patch / g) % g, px = patch % g;
    const int kk = 3 * ps * ps;
    const long img_off = (long)img * 3 * res * res;
    for (int k = threadIdx.x; k < kpad; k += 256) {
        float v = 0.f;
        if (k < kk) {
            const int c = k / (ps * ps), ky = (k / ps) % ps, kx = k % ps;
            const long idx = img_off + ((long)c * res + (py * ps + ky)) * res + (px * ps + kx);
            if (U8) {
                const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), std = c == 0 ? s0 : (c == 1 ? s1 : s2);
                v = ((float)((const unsigned char*)images)[idx] / 255.0f - mean) / std;
            } else {
                v = ((const float*)images)[idx];
            }
        }
        out[(long)patch * kpad + k] = Half16<T>::from(v);
    }
}

// uint8 frames, one workgroup per (image, row of patches): the 3 x ps x res bytes of the band are copied to LDS with coalesced
// 4-byte loads (a patch row is only ps bytes long: read from memory one output element at a time, the kernel above ran at
// 1 TB/s), then every thread builds 8 consecutive output columns at a time (16-byte stores).  The index split k -> (c, ky, kx)
// is done once per 8 columns.  Same arithmetic per element as above (bit-identical results).
template <typename T>
__global__ __launch_bounds__(256) void im2col_u8_band(const unsigned char* images, T* out, int res, int ps, int g, int kpad,
                                                     float m0, float m1, float m2, float s0, float s1, float s2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char band[];      // [3][ps][res]
    typedef typename Half16<T>::v8 v8;
    const int img = blockIdx.x / g, py = blockIdx.x % g;
    const int rowb = ps * res, pp = ps * ps, kk = 3 * pp;
    for (int c = 0; c < 3; ++c) {
        const unsigned char* src = images + ((long)(img * 3 + c) * res + (long)py * ps) * res;
        for (int i = threadIdx.x * 4; i < rowb; i += 1024) *(unsigned*)(band + c * rowb + i) = *(const unsigned*)(src + i);
    }
    __syncthreads();
    const int cpr = kpad >> 3;                          // 8-column chunks per output row
    for (int ch = threadIdx.x; ch < g * cpr; ch += 256) {
        const int px = ch / cpr, k0 = (ch - px * cpr) * 8;
        int c = k0 / pp, rem = k0 - c * pp;
        int ky = rem / ps, kx = rem - ky * ps;
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = 0.f;
            if (k0 + e < kk) {
                const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), std = c == 0 ? s0 : (c == 1 ? s1 : s2);
                v = ((float)band[c * rowb + ky * res + px * ps + kx] / 255.0f - mean) / std;
            }
            o[e] = Half16<T>::from(v);
            if (++kx == ps) { kx = 0; if (++ky == ps) { ky = 0; ++c; } }
        }
        *(v8*)(out + ((long)(img * g + py) * g + px) * kpad + k0) = o;
    }
}

// ---------------------------------------------------------------- y = x / ||x||_2 per row (fp32 in/out)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_kernel(const T* x, long ldx, float* y, long ldy, int rows, int width) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* src = x + (long)row * ldx;
    float sq = 0.f;
    for (int i = lane; i < width; i += 64) { const float v = (float)src[i]; sq += v * v; }
    const float rn = 1.0f / sqrtf(wave_sum(sq));
    for (int i = lane; i < width; i += 64) y[(long)row * ldy + i] = (float)src[i] * rn;
}

// ---------------------------------------------------------------- y = act(x Wt + b), fp32, small shapes
// Wt is [K][N] (transposed weight).  One workgroup handles 8 rows x 64 columns; x rows are staged in LDS.  These calls
// have one to a few rows (the proprioception MLP of a rollout step), so the time is the chain of dependent weight loads:
// each of the 4 waves sums a quarter of K for the 64 columns (coalesced 256-byte weight reads, 16 loads in flight per
// lane), the four partial sums are added in wave order through LDS (deterministic).
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* x, long ldx, const float* wt, const float* bias,
                                                         float* y, long ldy, int M, int N, int K, int act) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [8][K], then [4][8][64] partial sums
    float* red = xs + 8 * K;
    const int r0 = blockIdx.x * 8, n0 = blockIdx.y * 64;
    const int nr = min(8, M - r0);
    for (int i = threadIdx.x; i < 8 * K; i += 256) {
        const int r = i / K, k = i % K;
        xs[i] = (r < nr) ? x[(long)(r0 + r) * ldx + k] : 0.f;
    }
    __syncthreads();
    const int c = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int n = min(n0 + c, N - 1);                         // columns past N repeat the last one and are not stored
    const int kq = (K + 3) / 4;
    const int k0 = ks * kq, k1 = min(K, k0 + kq);
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
#pragma unroll 16
    for (int k = k0; k < k1; ++k) {
        const float w = wt[(long)k * N + n];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] += xs[r * K + k] * w;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) red[(ks * 8 + r) * 64 + c] = acc[r];
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * 64; i += 256) {
        const int r = i >> 6, cc = i & 63;
        if (r >= nr || n0 + cc >= N) continue;
        float v = bias ? bias[n0 + cc] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) v += red[(q * 8 + r) * 64 + cc];
        if (act == OTTER_EPI_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == OTTER_EPI_ACT_GELU) v = gelu_erf(v);
        else if (act == OTTER_EPI_ACT_QUICKGELU) v = quick_gelu(v);
        y[(long)(r0 + r) * ldy + n0 + cc] = v;
    }
}

// ---------------------------------------------------------------- fp32 -> {bf16, fp16, fp32} row copy with zero-padding and row map
// dst[r][c] = T(src[row_of(r)][c]) for c < cols, 0 for cols <= c < cols_pad.
//   row_of(r) = r / rep                       (idx == null)
//   row_of(r) = r * group + idx[r] + bias     (idx != null: pick one row out of each group of `group` rows)
template <typename T> __device__ __forceinline__ T conv_out(float v) { return Half16<T>::from(v); }
template <> __device__ __forceinline__ float conv_out<float>(float v) { return v; }

template <typename T>
__global__ __launch_bounds__(256) void convert_rows_kernel(const float* src, long lds_, T* dst, long ldd, int rows, int cols,
                                                           int cols_pad, int rep, const int* idx, int group, int bias) {
    const int row = blockIdx.x;
    const long srow = idx ? (long)row * group + min(max(idx[row] + bias, 0), group - 1) : (long)(row / rep);
    for (int c = threadIdx.x; c < cols_pad; c += 256)
        dst[(long)row * ldd + c] = conv_out<T>(c < cols ? src[srow * lds_ + c] : 0.f);
}

// dst[r][0:cols] = src[r / rep][0:cols]   (fp32)
__global__ __launch_bounds__(256) void repeat_rows_kernel(const float* src, long lds_, float* dst, long ldd, int rows, int cols, int rep) {
    const int row = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += 256) dst[(long)row * ldd + c] = src[(long)(row / rep) * lds_ + c];
}

// out[b*tq + i][:] = window[b][pos0 + i][:] + pe[pos0 + i][:]
__global__ __launch_bounds__(256) void add_pe_kernel(const float* window, const float* pe, float* out, int S, int tq, int pos0, int dim) {
    const int row = blockIdx.x, b = row / tq, i = row % tq;
    const float* src = window + ((long)b * S + pos0 + i) * dim;
    const float* p = pe + (long)(pos0 + i) * dim;
    for (int c = threadIdx.x; c < dim; c += 256) out[(long)row * dim + c] = src[c] + p[c];
}

// Ragged form for episodes that are at different positions: row i belongs to episode idx[i], whose newest input sits at window
// position lens[idx[i]] - 1 (lens: the lengths AFTER the append of this step).
__global__ __launch_bounds__(256) void add_pe_rows_kernel(const float* window, const float* pe, float* out, const int* idx, const int* lens,
                                                          int n_episodes, int S, int dim) {
    const int row = blockIdx.x;
    const int e = min(max(idx[row], 0), n_episodes - 1);
    const int pos = min(max(lens[e] - 1, 0), S - 1);
    const float* src = window + ((long)e * S + pos) * dim;
    const float* p = pe + (long)pos * dim;
    for (int c = threadIdx.x; c < dim; c += 256) out[(long)row * dim + c] = src[c] + p[c];
}

// Rollout window of policy inputs, one per episode (the reference's cached_transformer_input):
//   len < S : dst = src with `fresh` written at position len      (new length len+1)
//   len = S : dst = src shifted left by one, `fresh` at S-1       (length stays S)
// lens_in/lens_out are per-episode; src and dst are distinct buffers (ping-pong), rows beyond the length are zero.
__global__ __launch_bounds__(256) void window_append_kernel(const float* src, const float* fresh, float* dst, const int* lens_in,
                                                            int* lens_out, int S, int dim) {
    const int b = blockIdx.x / S, j = blockIdx.x % S;
    const int len = min(max(lens_in[b], 0), S);
    const bool full = (len == S);
    const float* from = nullptr;
    if (full) from = (j < S - 1) ? src + ((long)b * S + j + 1) * dim : fresh + (long)b * dim;
    else if (j < len) from = src + ((long)b * S + j) * dim;
    else if (j == len) from = fresh + (long)b * dim;
    for (int c = threadIdx.x; c < dim; c += 256) dst[((long)b * S + j) * dim + c] = from ? from[c] : 0.f;
    if (j == 0 && threadIdx.x == 0) lens_out[b] = full ? S : len + 1;
}

}  // namespace

template <typename T>
static void launch_im2col(const void* images, int is_u8, void* out, int n_img, int res, int patch, int g, int kpad,
                          const float* mean3, const float* std3, hipStream_t s) {
    const int band_bytes = 3 * patch * res;
    if (is_u8 && res % 4 == 0 && band_bytes <= 64 * 1024 && ((uintptr_t)images & 3) == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL((im2col_u8_band<T>), dim3(n_img * g), dim3(256), band_bytes, s, (const unsigned char*)images, (T*)out, res, patch, g,
                           kpad, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    else if (is_u8)
        hipLaunchKernelGGL((im2col_kernel<true, T>), dim3(n_img * g * g), dim3(256), 0, s, images, (T*)out, res, patch, g, kpad,
                           mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    else
        hipLaunchKernelGGL((im2col_kernel<false, T>), dim3(n_img * g * g), dim3(256), 0, s, images, (T*)out, res, patch, g, kpad,
                           0.f, 0.f, 0.f, 1.f, 1.f, 1.f);
}

extern "C" int otter_im2col_patches(int out_dtype, const void* images, int is_u8, void* out, int n_img, int res, int patch, int kpad,
                                    const float* mean3, const float* std3, void* stream) {
    OTTER_REQUIRE(images && out, OTTER_ERR_INVALID_ARGUMENT, "otter_im2col_patches: null operand");
    if (n_img == 0) return OTTER_OK;
    OTTER_REQUIRE(n_img > 0 && res > 0 && patch > 0 && res % patch == 0, OTTER_ERR_UNSUPPORTED_SHAPE,
                  "otter_im2col_patches: res=%d must be a positive multiple of patch=%d", res, patch);
    OTTER_REQUIRE(kpad >= 3 * patch * patch && kpad % 64 == 0, OTTER_ERR_UNSUPPORTED_SHAPE,
                  "otter_im2col_patches: kpad=%d must be a multiple of 64 and >= %d", kpad, 3 * patch * patch);
    const int g = res / patch;
    OTTER_REQUIRE((long)n_img * g * g < (1L << 31), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_im2col_patches: grid too large");
    OTTER_REQUIRE(out_dtype == DT_BF16 || out_dtype == DT_F16, OTTER_ERR_INVALID_ARGUMENT, "otter_im2col_patches: out_dtype %d (bf16 or fp16)", out_dtype);
    OTTER_REQUIRE(!is_u8 || (mean3 && std3), OTTER_ERR_INVALID_ARGUMENT, "otter_im2col_patches: uint8 input needs mean/std (host pointers)");
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == DT_BF16) launch_im2col<__bf16>(images, is_u8, out, n_img, res, patch, g, kpad, mean3, std3, s);
    else launch_im2col<_Float16>(images, is_u8, out, n_img, res, patch, g, kpad, mean3, std3, s);
    OTTER_CHECK_LAUNCH("otter_im2col_patches");
    return OTTER_OK;
}

extern "C" int otter_l2norm_rows(int x_dtype, const void* x, long ldx, float* y, long ldy, int rows, int width, void* stream) {
    OTTER_REQUIRE(x && y, OTTER_ERR_INVALID_ARGUMENT, "otter_l2norm_rows: null operand");
    if (rows == 0) return OTTER_OK;
    OTTER_REQUIRE(rows > 0 && width > 0 && ldx >= width && ldy >= width, OTTER_ERR_INVALID_ARGUMENT, "otter_l2norm_rows: bad sizes");
    OTTER_REQUIRE(x_dtype == DT_BF16 || x_dtype == DT_F16 || x_dtype == DT_F32, OTTER_ERR_INVALID_ARGUMENT, "otter_l2norm_rows: unknown dtype %d", x_dtype);
    if (x_dtype == DT_BF16)
        hipLaunchKernelGGL(l2norm_kernel<__bf16>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, ldx, y, ldy, rows, width);
    else if (x_dtype == DT_F16)
        hipLaunchKernelGGL(l2norm_kernel<_Float16>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, ldx, y, ldy, rows, width);
    else
        hipLaunchKernelGGL(l2norm_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, y, ldy, rows, width);
    OTTER_CHECK_LAUNCH("otter_l2norm_rows");
    return OTTER_OK;
}

extern "C" int otter_linear_f32(const float* x, long ldx, const float* wt, const float* bias, float* y, long ldy,
                                int M, int N, int K, int act, void* stream) {
    OTTER_REQUIRE(x && wt && y, OTTER_ERR_INVALID_ARGUMENT, "otter_linear_f32: null operand");
    if (M == 0) return OTTER_OK;
    OTTER_REQUIRE(M > 0 && N > 0 && K > 0 && K <= 2048, OTTER_ERR_UNSUPPORTED_SHAPE, "otter_linear_f32: M=%d N=%d K=%d (K <= 2048)", M, N, K);
    OTTER_REQUIRE(ldx >= K && ldy >= N, OTTER_ERR_INVALID_ARGUMENT, "otter_linear_f32: leading dimensions too small");
    const size_t lds = sizeof(float) * (8 * (size_t)K + 4 * 8 * 64);
    if (lds > 64 * 1024)
        if (int rc = otter_reserve_lds((const void*)linear_f32_kernel, lds, "otter_linear_f32")) return rc;
    hipLaunchKernelGGL(linear_f32_kernel, dim3((M + 7) / 8, (N + 63) / 64), dim3(256), lds, (hipStream_t)stream,
                       x, ldx, wt, bias, y, ldy, M, N, K, act);
    OTTER_CHECK_LAUNCH("otter_linear_f32");
    return OTTER_OK;
}

extern "C" int otter_convert_rows(int out_dtype, const float* src, long ld_src, void* dst, long ld_dst, int rows, int cols, int cols_pad,
                                  int rep, const int* pick, int group, int pick_bias, void* stream) {
    OTTER_REQUIRE(src && dst, OTTER_ERR_INVALID_ARGUMENT, "otter_convert_rows: null operand");
    if (rows == 0) return OTTER_OK;
    OTTER_REQUIRE(rows > 0 && cols > 0 && cols_pad >= cols && ld_src >= cols && ld_dst >= cols_pad && rep >= 1, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_convert_rows: bad sizes");
    OTTER_REQUIRE(!pick || group >= 1, OTTER_ERR_INVALID_ARGUMENT, "otter_convert_rows: pick needs group >= 1");
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == DT_BF16)
        hipLaunchKernelGGL(convert_rows_kernel<__bf16>, dim3(rows), dim3(256), 0, s, src, ld_src, (__bf16*)dst, ld_dst, rows, cols, cols_pad, rep, pick, group, pick_bias);
    else if (out_dtype == DT_F16)
        hipLaunchKernelGGL(convert_rows_kernel<_Float16>, dim3(rows), dim3(256), 0, s, src, ld_src, (_Float16*)dst, ld_dst, rows, cols, cols_pad, rep, pick, group, pick_bias);
    else if (out_dtype == DT_F32)
        hipLaunchKernelGGL(convert_rows_kernel<float>, dim3(rows), dim3(256), 0, s, src, ld_src, (float*)dst, ld_dst, rows, cols, cols_pad, rep, pick, group, pick_bias);
    else {
        otter_set_error("otter_convert_rows: unknown out_dtype %d", out_dtype);
        return OTTER_ERR_INVALID_ARGUMENT;
    }
    OTTER_CHECK_LAUNCH("otter_convert_rows");
    return OTTER_OK;
}

extern "C" int otter_repeat_rows(const float* src, long ld_src, float* dst, long ld_dst, int rows, int cols, int rep, void* stream) {
    OTTER_REQUIRE(src && dst, OTTER_ERR_INVALID_ARGUMENT, "otter_repeat_rows: null operand");
    if (rows == 0) return OTTER_OK;
    OTTER_REQUIRE(rows > 0 && cols > 0 && rep >= 1 && ld_src >= cols && ld_dst >= cols, OTTER_ERR_INVALID_ARGUMENT, "otter_repeat_rows: bad sizes");
    hipLaunchKernelGGL(repeat_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, rows, cols, rep);
    OTTER_CHECK_LAUNCH("otter_repeat_rows");
    return OTTER_OK;
}

extern "C" int otter_add_pe(const float* window, const float* pe, float* out, int batch, int window_len, int n_query, int pos0,
                            int dim, void* stream) {
    OTTER_REQUIRE(window && pe && out, OTTER_ERR_INVALID_ARGUMENT, "otter_add_pe: null operand");
    if (batch == 0 || n_query == 0) return OTTER_OK;
    OTTER_REQUIRE(batch > 0 && n_query > 0 && pos0 >= 0 && pos0 + n_query <= window_len && dim > 0, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_add_pe: positions %d..%d outside the window of %d", pos0, pos0 + n_query - 1, window_len);
    hipLaunchKernelGGL(add_pe_kernel, dim3(batch * n_query), dim3(256), 0, (hipStream_t)stream, window, pe, out, window_len, n_query, pos0, dim);
    OTTER_CHECK_LAUNCH("otter_add_pe");
    return OTTER_OK;
}

extern "C" int otter_add_pe_rows(const float* window, const float* pe, float* out, const int* idx, const int* lens, int n_rows,
                                 int n_episodes, int window_len, int dim, void* stream) {
    OTTER_REQUIRE(window && pe && out && idx && lens, OTTER_ERR_INVALID_ARGUMENT, "otter_add_pe_rows: null operand");
    if (n_rows == 0) return OTTER_OK;
    OTTER_REQUIRE(n_rows > 0 && n_episodes > 0 && window_len > 0 && dim > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_add_pe_rows: bad sizes");
    hipLaunchKernelGGL(add_pe_rows_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, window, pe, out, idx, lens, n_episodes, window_len, dim);
    OTTER_CHECK_LAUNCH("otter_add_pe_rows");
    return OTTER_OK;
}

extern "C" int otter_window_append(const float* src, const float* fresh, float* dst, const int* lens_in, int* lens_out,
                                   int batch, int window_len, int dim, void* stream) {
    OTTER_REQUIRE(src && fresh && dst && lens_in && lens_out, OTTER_ERR_INVALID_ARGUMENT, "otter_window_append: null operand");
    OTTER_REQUIRE(src != dst && lens_in != lens_out, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_window_append: src/dst and lens_in/lens_out must be different buffers");
    if (batch == 0) return OTTER_OK;
    OTTER_REQUIRE(batch > 0 && window_len > 0 && dim > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_window_append: bad sizes");
    hipLaunchKernelGGL(window_append_kernel, dim3(batch * window_len), dim3(256), 0, (hipStream_t)stream, src, fresh, dst, lens_in,
                       lens_out, window_len, dim);
    OTTER_CHECK_LAUNCH("otter_window_append");
    return OTTER_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Centre-crop + bicubic resize of uint8 RGB camera frames, bit-identical to PIL's 8-bit resampler (what
// otter_interface.py:98-108 does on the host with Image.resize(..., BICUBIC)): two separable passes, horizontal
// first, each output = clip8((2^21 + sum_k coeff[k] * pixel[k]) >> 22) with the intermediate image rounded to
// uint8.  The integer coefficient tables (22 fractional bits) come from the host, computed as PIL computes them.
namespace {

__device__ __forceinline__ unsigned char clip8_fixed(int acc) {
    const int v = acc >> 22;                                  // arithmetic shift
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// src: [n][src_h][src_w][3] interleaved, the crop starts at (y0, x0) and is crop_h x crop_w; tmp: [n][crop_h][S][3]
__global__ void resize_h_kernel(const unsigned char* src, unsigned char* tmp, const int* bounds, const int* coef, int ksize,
                                int n, int src_h, int src_w, int y0, int x0, int crop_h, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n * crop_h * S * 3;
    if (i >= total) return;
    const int c = (int)(i % 3), xx = (int)((i / 3) % S), yy = (int)((i / (3L * S)) % crop_h), img = (int)(i / (3L * S * crop_h));
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const unsigned char* row = src + (((long)img * src_h + y0 + yy) * src_w + x0 + xmin) * 3 + c;
    int acc = 1 << 21;
    for (int k = 0; k < cnt; ++k) acc += (int)row[3 * k] * coef[xx * ksize + k];
    tmp[i] = clip8_fixed(acc);
}

// tmp: [n][crop_h][S][3] -> out: [n][3][S][S] planar (the layout the patch-extraction kernel reads)
__global__ void resize_v_kernel(const unsigned char* tmp, unsigned char* out, const int* bounds, const int* coef, int ksize,
                                int n, int crop_h, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n * 3 * S * S;
    if (i >= total) return;
    const int xx = (int)(i % S), yy = (int)((i / S) % S), c = (int)((i / ((long)S * S)) % 3), img = (int)(i / (3L * S * S));
    const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
    const unsigned char* col = tmp + (((long)img * crop_h + ymin) * S + xx) * 3 + c;
    int acc = 1 << 21;
    for (int k = 0; k < cnt; ++k) acc += (int)col[(long)k * S * 3] * coef[yy * ksize + k];
    out[i] = clip8_fixed(acc);
}

}  // namespace

extern "C" int otter_resize_bicubic_u8(const unsigned char* src, int n, int src_h, int src_w, int y0, int x0, int crop_h, int crop_w,
                                       const int* h_bounds, const int* h_coef, int h_ksize, const int* v_bounds, const int* v_coef,
                                       int v_ksize, unsigned char* tmp, unsigned char* out, int out_size, void* stream) {
    OTTER_REQUIRE(src && h_bounds && h_coef && v_bounds && v_coef && tmp && out, OTTER_ERR_INVALID_ARGUMENT, "otter_resize_bicubic_u8: null operand");
    if (n == 0) return OTTER_OK;
    OTTER_REQUIRE(n > 0 && out_size > 0 && crop_h > 0 && crop_w > 0 && y0 >= 0 && x0 >= 0 && y0 + crop_h <= src_h && x0 + crop_w <= src_w &&
                  h_ksize > 0 && v_ksize > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_resize_bicubic_u8: crop (%d,%d)+(%d,%d) outside %dx%d or bad sizes",
                  y0, x0, crop_h, crop_w, src_h, src_w);
    hipStream_t s = (hipStream_t)stream;
    const long th = (long)n * crop_h * out_size * 3, tv = (long)n * 3 * out_size * out_size;
    OTTER_REQUIRE((th + 255) / 256 < (1L << 31), OTTER_ERR_UNSUPPORTED_SHAPE, "otter_resize_bicubic_u8: batch too large");
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, src, tmp, h_bounds, h_coef, h_ksize, n, src_h, src_w,
                       y0, x0, crop_h, out_size);
    OTTER_CHECK_LAUNCH("otter_resize_bicubic_u8 (horizontal)");
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((tv + 255) / 256)), dim3(256), 0, s, tmp, out, v_bounds, v_coef, v_ksize, n, crop_h, out_size);
    OTTER_CHECK_LAUNCH("otter_resize_bicubic_u8 (vertical)");
    return OTTER_OK;
}
