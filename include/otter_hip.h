/* C ABI of libotter_hip.so: the MI355X (gfx950) kernels behind OTTER's observation -> action-chunk path.
 *
 * The reference (jesbu1/otter) is pure Python/PyTorch and has no native ABI of its own; every entry point
 * below replaces a stock-torch op sequence of the reference, cited as file:line under the reference tree.
 *
 * Conventions
 *   - plain pointers and sizes only; all data pointers are DEVICE pointers unless a comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work, they
 *     never allocate, free or synchronise, so they can be captured into a HIP graph;
 *   - the caller owns every buffer; leading dimensions (`ld*`) are in ELEMENTS of the buffer's type;
 *   - bf16 buffers are raw 16-bit brain-float; "fp32" buffers are IEEE float;
 *   - return value: 0 on success, otherwise an OTTER_ERR_* code; otter_last_error() returns a
 *     thread-local message for the last failure.  No exception crosses this boundary.
 */
#ifndef OTTER_HIP_H
#define OTTER_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define OTTER_HIP_ABI_VERSION 2

/* element type codes */
#define OTTER_DT_BF16 0
#define OTTER_DT_F16 1
#define OTTER_DT_F32 2

/* epilogue word of otter_gemm: activation | flags */
#define OTTER_EPI_ACT_NONE 0
#define OTTER_EPI_ACT_QUICKGELU 1 /* x * sigmoid(1.702 x): OpenAI CLIP MLP */
#define OTTER_EPI_ACT_GELU 2      /* exact erf GELU: policy MLP (transformer.py:116), OpenCLIP towers */
#define OTTER_EPI_ACT_RELU 3      /* action head / proprio MLP (models.py:42-47,58-62) */
#define OTTER_EPI_ACT_MASK 3
#define OTTER_EPI_OUT_F32 4       /* write fp32 instead of the 16-bit operand type (always on for fp32 operands) */
#define OTTER_EPI_RESIDUAL 8      /* out = acc + bias + residual (fp32), requires OUT_F32 and ACT_NONE */
#define OTTER_EPI_TILE_128 16     /* tuning knob: always use the 128x128-tile kernel (default picks 256x256 for big shapes) */

int otter_hip_abi_version(void);
const char* otter_last_error(void);

/* C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N]) on MFMA, fp32 accumulation.
 * dtype: operand type of A and W.  OTTER_DT_BF16 / OTTER_DT_F16 (v_mfma_f32_16x16x32_*; the CLIP towers) need
 * K % 64 == 0; OTTER_DT_F32 (v_mfma_f32_16x16x4_f32, an exact fp32 fma chain; the small policy / pooling / head
 * layers) needs K % 32 == 0.  A: row-major, lda >= K.  W: [n_pad][K] in torch Linear layout (out, in) with
 * n_pad >= N rounded up to a multiple of 128 (extra rows may hold anything finite).  N % 4 == 0.
 * C: the operand's 16-bit type, or fp32 with OTTER_EPI_OUT_F32 / fp32 operands.
 * Replaces every nn.Linear / `x @ proj` / in_proj of the path: CLIP resblocks (clip.encode_image / encode_text
 * called at otter/policy/otter.py:189-190), `@ visual.proj` / `@ text_projection` (otter.py:194,212),
 * policy query/key/value/out/mlp (otter/policy/transformer.py:64-69,114-118), input_projection (otter.py:262),
 * action head (otter/policy/models.py:58-62), pooling k/v projections (transformer.py:165-171). */
int otter_gemm(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias, const float* residual,
               long ldr, void* C, long ldc, int M, int N, int K, int epilogue, void* stream);

/* y[r,:] = LayerNorm(x[r,:]) * gamma + beta  (fp32 in; out_dtype = OTTER_DT_*; gamma/beta NULL = no affine).
 * Replaces ln_1/ln_2/ln_post/ln_final of CLIP (otter.py:194,209), pre_ln/post_ln (transformer.py:111-112,133,138),
 * kv_layer_norm (transformer.py:190). */
int otter_layernorm(const float* x, long ldx, const float* gamma, const float* beta, void* y, long ldy, int rows,
                    int width, float eps, int out_dtype, void* stream);

/* ViT input: x[i, t, :] = ln_pre((t == 0 ? class_embedding : patch_tok[i, t-1, :]) + positional_embedding[t, :]).
 * Replaces the class-token concat + positional add + ln_pre inside clip's VisionTransformer.forward
 * (reached from otter.py:190). */
int otter_vit_embed_lnpre(const float* patch_tok, long ld_tok, const float* class_embedding, const float* pos_embedding,
                          const float* gamma, const float* beta, float* x, long ldx, int n_img, int n_patches, int width,
                          float eps, void* stream);

/* Text-tower input: x[b*take + t, :] = token_embedding[ids[b*ctx + t]] + positional_embedding[t], t < take.
 * (clip.encode_text reached from otter.py:189; only the first `take` positions are needed, otter.py:193.) */
int otter_text_embed(const int* ids, int ctx, int take, const float* token_embedding, int vocab, const float* pos_embedding,
                     float* x, long ldx, int batch, int width, void* stream);

/* Patch extraction for the stride=kernel conv1 of the ViT as a GEMM operand:
 * out[(i*g*g + py*g + px), c*p*p + ky*p + kx] = pixel(i, c, py*p+ky, px*p+kx), zero-padded to kpad columns
 * (out_dtype bf16 or fp16).
 * is_u8 != 0: images are uint8 (N,3,res,res) and the CLIP normalisation /255, (x-mean)/std of
 * otter/policy/vision_tf.py:35-45 is fused (mean3/std3: HOST pointers to 3 floats);
 * is_u8 == 0: images are fp32, already normalised. */
int otter_im2col_patches(int out_dtype, const void* images, int is_u8, void* out, int n_img, int res, int patch, int kpad,
                         const float* mean3, const float* std3, void* stream);

/* Multi-head self-attention with head_dim 64 over packed q|k|v rows (row = seq*L + t, width 3*heads*64);
 * dtype bf16 or fp16 (qkv and out).
 * L <= 288.  causal != 0 applies the text tower's mask.  Replaces nn.MultiheadAttention inside the CLIP
 * resblocks, including the hooked last vision block (otter.py:141-143). */
int otter_attn_fwd_d64(int dtype, const void* qkv, long ld, void* out, long ldo, int n_seq, int L, int heads, int causal, void* stream);

/* Causal attention of the policy transformer over a GPU-resident history (otter/policy/transformer.py:76-104,
 * cache semantics otter.py:153-175,319-329).  qkv: fp32 rows (b*window + t) of width 3*heads*head_dim that double
 * as the K/V history; queries are the n_query rows at positions pos0 .. pos0+n_query-1, each attending keys
 * 0 .. its own position.  out: fp32 rows (b*n_query + i), width heads*head_dim.  window <= 64. */
int otter_attn_policy(const float* qkv, long ld, float* out, long ldo, int batch, int window, int n_query, int pos0,
                      int heads, int head_dim, void* stream);

/* Fused text-aware visual extraction (otter/policy/models.py:22-36 plus the L2 normalisations of
 * otter.py:195,214): for each (observation m, camera c)
 *   out[c, m, k, :] = sum_p softmax_p( <t^_k, x^_p> / clamp(temperature[c],0,100) ) * (x^_p + pos_emb[c, p, :])
 * dtype: bf16 or fp16, the type of `patches` and `posT`.
 * patches: un-normalised projected patch tokens; unit u = m*cams + c is the row-major [P][D] block at
 *   patches + u*unit_stride (unit_stride in elements, >= P*D; lets the caller skip the class-token row);
 * text: fp32 [n_obs/t_rep][K][D] un-normalised projected text tokens (observation m uses row m / t_rep);
 * posT: [cams][D][P] from otter_pos_emb_transpose; out: fp32 [cams][n_obs][K][D].  K <= 16, P % 16 == 0. */
int otter_text_aware_pool_fwd(int dtype, const void* patches, long unit_stride, const float* text, const void* posT, const float* temperature,
                              float* out, int n_obs, int cams, int P, int D, int K, int t_rep, void* stream);
int otter_pos_emb_transpose(int dtype, const float* pos /* [P][D] fp32 */, void* posT /* [D][P] bf16|fp16 */, int P, int D, void* stream);

/* Query side of AttentionPooling (otter/policy/transformer.py:162-232) after the k/v projections.
 * All weights fp32, matrices transposed to [in][out]. */
typedef struct {
    const float* q_ln_w; const float* q_ln_b; /* q_layer_norm [d] */
    const float* wq_t;   const float* bq;     /* q_proj_weight^T [d][d], in_proj_bias[0:d] */
    const float* wo_t;   const float* bo;     /* out_proj.weight^T [d][d], out_proj.bias */
    const float* ln_w;   const float* ln_b;   /* layer_norm [d] */
    const float* fc1_t;  const float* fc1_b;  /* mlp.fc1.weight^T [d][hidden], bias */
    const float* fc2_t;  const float* fc2_b;  /* mlp.fc2.weight^T [hidden][d], bias */
} otter_readout_block;
/* blocks: HOST array of n_blocks structs holding device pointers.  kv: fp32 rows (m*nk + k); block b's keys at
 * columns [b*2d, b*2d+d), values at [b*2d+d, (b+1)*2d).  mask: [n_obs][nk] bytes, non-zero = ignore key, or NULL.
 * out: fp32 [n_obs][readouts*d]. */
int otter_readout_pool_fwd(const otter_readout_block* blocks, int n_blocks, const float* query, const float* final_ln_w,
                           const float* final_ln_b, const float* kv, long ldkv, const unsigned char* mask, float* out,
                           long ldo, int n_obs, int nk, int readouts, int d, int heads, int hidden, void* stream);

/* y = x / ||x||_2 per row (otter.py:195,214); x has type x_dtype (OTTER_DT_*), y is fp32. */
int otter_l2norm_rows(int x_dtype, const void* x, long ldx, float* y, long ldy, int rows, int width, void* stream);

/* y = act(x . wt + bias) in fp32 for small layers; wt is the transposed weight [K][N]
 * (ProprioceptionEncoder, otter/policy/models.py:42-51). act: OTTER_EPI_ACT_*. */
int otter_linear_f32(const float* x, long ldx, const float* wt, const float* bias, float* y, long ldy, int M, int N, int K,
                     int act, void* stream);

/* dst[r, c] = T(src[row_of(r), c]) for c < cols, 0 for cols <= c < cols_pad; T = out_dtype (OTTER_DT_*), src fp32.
 * pick == NULL: row_of(r) = r / rep.  pick != NULL: row_of(r) = r*group + clamp(pick[r] + pick_bias, 0, group-1)
 * (select one row per group, e.g. the last valid step of each episode's window: pick = lengths, pick_bias = -1). */
int otter_convert_rows(int out_dtype, const float* src, long ld_src, void* dst, long ld_dst, int rows, int cols, int cols_pad,
                       int rep, const int* pick, int group, int pick_bias, void* stream);

/* dst[r, 0:cols] = src[r / rep, 0:cols] (fp32): repeat_interleave of otter.py:200 for pooled text tokens. */
int otter_repeat_rows(const float* src, long ld_src, float* dst, long ld_dst, int rows, int cols, int rep, void* stream);

/* out[b*n_query + i, :] = window[b, pos0+i, :] + pe[pos0+i, :]  (SinusoidalPositionalEmbedding.forward,
 * otter/policy/transformer.py:23-24, applied to the cached window at otter.py:325). */
int otter_add_pe(const float* window, const float* pe, float* out, int batch, int window_len, int n_query, int pos0, int dim,
                 void* stream);

/* GPU-resident per-episode window of policy inputs (OTTER.update_cache, otter.py:158-175): appends `fresh[b]` to
 * episode b, dropping the oldest entry once the window is full.  src/dst and lens_in/lens_out are distinct
 * (ping-pong) buffers; window rows beyond an episode's length are zero. */
int otter_window_append(const float* src, const float* fresh, float* dst, const int* lens_in, int* lens_out, int batch,
                        int window_len, int dim, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OTTER_HIP_H */
