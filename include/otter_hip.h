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

#define OTTER_HIP_ABI_VERSION 14

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
#define OTTER_EPI_RESIDUAL 8      /* out = acc + bias + residual; residual has the OUTPUT's type (fp32 with OUT_F32, else the operand type); ACT_NONE */
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
int otter_gemm(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias, const void* residual,
               long ldr, void* C, long ldc, int M, int N, int K, int epilogue, void* stream);

/* LayerNorm folded into the GEMM that consumes it (ln_1 -> in_proj and ln_2 -> c_fc of a CLIP resblock, reached from
 * otter.py:189-190): with W' = W * gamma (column-wise), col_sum[n] = sum_k W'[n,k] (of the rounded 16-bit W') and
 * bias' = bias + W . beta,
 *   C[m,n] = act( rstd[m] * (A[m,:] . W'[n,:] - mean[m] * col_sum[n]) + bias'[n] )  ==  act(LayerNorm(A)[m,:] . W[n,:] + bias[n])
 * so the GEMM reads the residual stream itself and the normalised copy is never written.  row_stats: [M][2] =
 * (mean, rstd) from otter_row_stats; the other arguments as otter_gemm (no residual epilogue). */
int otter_gemm_ln(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias, const float* row_stats,
                  const float* col_sum, void* C, long ldc, int M, int N, int K, int epilogue, void* stream);
/* otter_gemm / otter_gemm_ln with a caller-provided scratch buffer (row_stats / col_sum NULL = no folded LayerNorm).
 * With it few-tile problems (a rollout step of one episode) and the < 256 remainder rows of a big 16-bit problem (the
 * CLIP towers have 32896 = 128 x 257 token rows) are
 * multiplied split-K over the whole chip: fp32 partial products go to `workspace`, a second launch sums them and applies
 * the epilogue.  workspace: >= otter_gemm_workspace_floats(N) floats, 16-byte aligned, private to the stream for the
 * duration of the call; NULL = as otter_gemm.  Results are deterministic (fixed summation order). */
long otter_gemm_workspace_floats(int N);
int otter_gemm_ws(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias, const void* residual,
                  long ldr, const float* row_stats, const float* col_sum, void* C, long ldc, int M, int N, int K, int epilogue,
                  float* workspace, long workspace_floats, void* stream);
/* otter_gemm_ws for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): the split-K scratch is a static device array inside the library (33.4 MB per GPU,
 * held until the process ends), so no workspace is passed and the call allocates nothing.  No counterpart in the
 * reference.  N <= otter_gemm_static_ws_max_n() (4096), larger N is rejected with OTTER_ERR_UNSUPPORTED_SHAPE; results
 * are bit-identical to otter_gemm_ws with a workspace of otter_gemm_workspace_floats(N) floats.  Every call on one GPU
 * shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race; serialise them.
 * Set-up: the array's device address is looked up once per GPU (hipGetSymbolAddress, which may load the code object and
 * is therefore neither allocation-free nor legal under stream capture).  otter_gemm_static_ws_init() does that look-up for
 * the CURRENT device; call it once per GPU before the first capture.  Without it the first otter_gemm_static_ws call on a
 * GPU does the look-up itself -- and refuses (OTTER_ERR_LAUNCH) if its stream is being captured at that moment.  The
 * stream must belong to the current device (checked). */
long otter_gemm_static_ws_max_n(void);
int otter_gemm_static_ws_init(void);
int otter_gemm_static_ws(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias, const void* residual,
                         long ldr, const float* row_stats, const float* col_sum, void* C, long ldc, int M, int N, int K,
                         int epilogue, void* stream);
/* The residual product of a CLIP resblock together with the statistics of the LayerNorm that reads its result
 * (`x = x + attn(ln_1(x))` followed by ln_2, `x = x + mlp(ln_2(x))` followed by the next block's ln_1: the resblocks reached
 * from otter.py:189-190): x[M][N] += A[M][K] W[N][K]^T + bias in place on the 16-bit residual stream, then
 * stats[r] = (mean, rstd) of the updated row r.  Bit for bit otter_gemm_ws with OTTER_EPI_RESIDUAL followed by
 * otter_row_stats; when the < 256 remainder rows are multiplied split-K their second pass shares one launch with the
 * statistics of all rows.  N <= 1024, N % 8 == 0; workspace as otter_gemm_ws (NULL: two plain launches). */
int otter_gemm_residual_stats(int dtype, const void* A, long lda, const void* W, int n_pad, const float* bias, void* x, long ldx,
                              int M, int N, int K, float* stats, float eps, float* workspace, long workspace_floats, void* stream);
/* stats[r] = (mean, 1/sqrt(var + eps)) of row r (two-pass, fp32), x fp32 or 16-bit. */
int otter_row_stats(const void* x, int in_dtype, long ldx, float* stats, int rows, int width, float eps, void* stream);

/* y[r,:] = LayerNorm(x[r,:]) * gamma + beta  (in_dtype / out_dtype = OTTER_DT_*: x may be the fp32 or the 16-bit
 * residual stream; statistics in fp32; gamma/beta NULL = no affine).
 * Replaces ln_1/ln_2/ln_post/ln_final of CLIP (otter.py:194,209), pre_ln/post_ln (transformer.py:111-112,133,138),
 * kv_layer_norm (transformer.py:190). */
int otter_layernorm(const void* x, int in_dtype, long ldx, const float* gamma, const float* beta, void* y, long ldy, int rows,
                    int width, float eps, int out_dtype, void* stream);

/* ViT input: x[i, t, :] = ln_pre((t == 0 ? class_embedding : patch_tok[i, t-1, :]) + positional_embedding[t, :]).
 * Replaces the class-token concat + positional add + ln_pre inside clip's VisionTransformer.forward
 * (reached from otter.py:190). */
int otter_vit_embed_lnpre(const float* patch_tok, long ld_tok, const float* class_embedding, const float* pos_embedding,
                          const float* gamma, const float* beta, void* x /* out_dtype */, int out_dtype, long ldx, int n_img,
                          int n_patches, int width, float eps, void* stream);

/* Text-tower input: x[b*take + t, :] = token_embedding[ids[b*ctx + t]] + positional_embedding[t], t < take.
 * (clip.encode_text reached from otter.py:189; only the first `take` positions are needed, otter.py:193.) */
int otter_text_embed(const int* ids, int ctx, int take, const float* token_embedding, int vocab, const float* pos_embedding,
                     void* x /* out_dtype */, int out_dtype, long ldx, int batch, int width, void* stream);

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
 * L <= 592 (K and V of one head stay in LDS).  causal != 0 applies the text tower's mask.  Replaces nn.MultiheadAttention inside the CLIP
 * resblocks, including the hooked last vision block (otter.py:141-143). */
int otter_attn_fwd_d64(int dtype, const void* qkv, long ld, void* out, long ldo, int n_seq, int L, int heads, int causal, void* stream);

/* Causal attention of the policy transformer over a GPU-resident history (otter/policy/transformer.py:76-104,
 * cache semantics otter.py:153-175,319-329).  qkv: fp32 rows (b*window + t) of width 3*heads*head_dim that double
 * as the K/V history; queries are the n_query rows at positions pos0 .. pos0+n_query-1, each attending keys
 * 0 .. its own position.  out: fp32 rows (b*n_query + i), width heads*head_dim.  window <= 64. */
int otter_attn_policy(const float* qkv, long ld, float* out, long ldo, int batch, int window, int n_query, int pos0,
                      int heads, int head_dim, void* stream);

/* The same attention for episodes that are at DIFFERENT positions of their windows (a server whose episodes start and
 * end independently; per-episode semantics of otter.py:153-175,319-329).  Row i is the newest step of episode idx[i], at
 * window position lens[idx[i]] - 1 (lens: per-episode lengths after this step's append, device int32).  fresh: fp32
 * [n_rows][ldf] q|k|v rows of the new steps; each is stored into its episode's row of `cache`
 * ([n_episodes][window][ld], the K/V history) and attends over that episode's earlier rows plus itself.
 * out: fp32 [n_rows][ldo].  window <= 64; idx values outside [0, n_episodes) and lengths outside [1, window] are clamped. */
int otter_attn_policy_rows(const float* fresh, long ldf, float* cache, long ld, float* out, long ldo, const int* idx,
                           const int* lens, int n_rows, int n_episodes, int window, int heads, int head_dim, void* stream);

/* Fused text-aware visual extraction (otter/policy/models.py:22-36 plus the L2 normalisations of
 * otter.py:195,214): for each (observation m, camera c)
 *   out[c, m, k, :] = sum_p softmax_p( <t^_k, x^_p> / clamp(temperature[c],0,100) ) * (x^_p + pos_emb[c, p, :])
 * dtype: bf16 or fp16, the type of `patches` and `posT`.
 * patches: un-normalised projected patch tokens; unit u = m*cams + c is the row-major [P][D] block at
 *   patches + u*unit_stride (unit_stride in elements, >= P*D; lets the caller skip the class-token row);
 * text: fp32 [n_obs/t_rep][K][D] un-normalised projected text tokens (observation m uses row m / t_rep);
 * pos_packed: [cams][otter_pos_emb_packed_elems(P, D)], every camera's pos_emb packed by otter_pos_emb_pack;
 * out: fp32 [cams][n_obs][K][D].  K <= 16, any P >= 1 (ViT-B/16: 196, ViT-L/14: 256, ViT-L/14@336px: 576),
 * D in {128, 512, 768, 1024}. */
int otter_text_aware_pool_fwd(int dtype, const void* patches, long unit_stride, const float* text, const void* pos_packed,
                              const float* temperature, float* out, int n_obs, int cams, int P, int D, int K, int t_rep,
                              void* stream);
/* The same kernel with 16-bit operands at its edges: io16 bit 0 = `text` holds the patches' 16-bit type instead of fp32
 * (the text tower's projection can write it directly), bit 1 = `out` is written in that type.  With both a unit moves
 * 256*768*2 + 2*15*768*2 = 439,296 bytes (the path's compulsory traffic at 16 bits) instead of 485,376.  io16 = 0 is
 * otter_text_aware_pool_fwd.  The arithmetic inside is unchanged (text rows are normalised in fp32, accumulation in fp32). */
int otter_text_aware_pool_fwd_io(int dtype, const void* patches, long unit_stride, const void* text, const void* pos_packed,
                                 const float* temperature, void* out, int n_obs, int cams, int P, int D, int K, int t_rep,
                                 int io16, void* stream);
/* pos_emb buffer (sincos_position_embedding, models.py:5-12,19-20) [P][D] fp32 -> 16-bit table in the order the
 * pooling kernel's lanes read it (16-byte coalesced loads); P is rounded up to a multiple of 16 with zero rows. */
long otter_pos_emb_packed_elems(int P, int D);
int otter_pos_emb_pack(int dtype, const float* pos /* [P][D] fp32 */, void* packed /* [otter_pos_emb_packed_elems] bf16|fp16 */,
                       int P, int D, void* stream);

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

/* The poolings of one encoder call in ONE launch: forward_encoder runs vision_poolings[c] for every camera and then
 * text_pooling (otter/policy/otter.py:247,254), each an AttentionPooling with weights and sizes of its own.  A pooling
 * is a chain of small dependent phases that takes the same ~40 us for 1 or for 256 observations, so the problems'
 * observations are concatenated into one grid.  Fields as the arguments of otter_readout_pool_fwd; a problem with
 * n_obs == 0 is skipped; at most 4 problems. */
typedef struct otter_readout_problem {
    const otter_readout_block* blocks; int n_blocks;    /* HOST array of structs holding device pointers */
    const float* query; const float* final_ln_w; const float* final_ln_b;
    const float* kv; long ldkv;
    const unsigned char* mask;
    float* out; long ldo;
    int n_obs, nk, readouts, d, heads, hidden;
} otter_readout_problem;
int otter_readout_pool_fwd_multi(const otter_readout_problem* problems, int n_problems, void* stream);

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

/* Ragged form: out[i, :] = window[e, p, :] + pe[p, :] with e = idx[i], p = lens[e] - 1 (the newest step of each listed
 * episode; transformer.py:23-24 applied per episode).  idx, lens: device int32. */
int otter_add_pe_rows(const float* window, const float* pe, float* out, const int* idx, const int* lens, int n_rows,
                      int n_episodes, int window_len, int dim, void* stream);

/* GPU-resident per-episode window of policy inputs (OTTER.update_cache, otter.py:158-175): appends `fresh[b]` to
 * episode b, dropping the oldest entry once the window is full.  src/dst and lens_in/lens_out are distinct
 * (ping-pong) buffers; window rows beyond an episode's length are zero. */
int otter_window_append(const float* src, const float* fresh, float* dst, const int* lens_in, int* lens_out, int batch,
                        int window_len, int dim, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Training step (fp32): forward pieces with saved pre-activations and hand-written backward of the trainable
 * modules (everything but CLIP, otter/policy/otter.py:75-128), loss, gradient statistics and AdamW.
 * The reference gets these from torch autograd + torch.optim (otter/util/engine.py:44-66, misc.py:329-348,
 * script/train.py:168-171).  Kernels that produce gradients ACCUMULATE (+=) into zero-initialised buffers.
 * --------------------------------------------------------------------------------------------------------------- */

/* C[M,N] (+)= op(A)[M,K] . op(B)[K,N] (+ bias[n]) with element strides: A(m,k) = A[m*sa_m + k*sa_k],
 * B(k,n) = B[k*sb_k + n*sb_n].  nn.Linear forward (transformer.py:38-50, models.py:42-66): A = x, B = W^T
 * (sb_k = 1, sb_n = ldw); input gradient dy.W: sb_k = ldw, sb_n = 1; weight gradient dy^T.x: sa_m = 1, sa_k = ld_dy.
 * a_rowsum (may be NULL): [M] floats that receive (+= with accumulate) sum_k A(m, k) -- with A = dy^T the bias gradient,
 * from the same launch as the weight gradient. */
int otter_sgemm_strided(const float* A, long sa_m, long sa_k, const float* B, long sb_k, long sb_n, const float* bias,
                        float* C, long ldc, int M, int N, int K, int accumulate, float* a_rowsum, void* stream);

/* y = act(z) and dz += dy * act'(z); act: OTTER_EPI_ACT_* (nn.GELU of transformer.py:75-79 / timm Mlp, nn.ReLU of models.py:42-66). */
int otter_act_fwd(const float* z, float* y, long n, int act, void* stream);
int otter_act_bwd(const float* z, const float* dy, float* dz, long n, int act, void* stream);

/* out[c] (+)= sum_r x[r, c] * (y ? y[r, c] : 1): bias gradients and gradients of parameters broadcast over rows. */
int otter_col_sum(const float* x, long ldx, const float* y, long ldy, int rows, int cols, float* out, int accumulate, void* stream);

/* nn.LayerNorm backward (transformer.py:62-63,176-178): dx += d loss / d x (dx may be NULL); with dgamma/dbeta
 * (both or neither) also dgamma += , dbeta += ; stats: workspace of 2*rows floats, required with dgamma.
 * gamma NULL = no affine. */
int otter_layernorm_bwd(const float* x, long ldx, const float* gamma, const float* dy, long ldy, float* dx, long lddx,
                        float* stats, float* dgamma, float* dbeta, int rows, int width, float eps, void* stream);

/* Small multi-head attention, fp32 (F.scaled_dot_product_attention of transformer.py:44-47 with is_causal, and
 * nn.MultiheadAttention with key_padding_mask of transformer.py:189-191).  Element (b, i, h, e) of q is
 * q[b*sq_b + i*sq_l + h*head_dim + e]; k and v share strides; sq_b = 0 broadcasts one query set over the batch.
 * key_mask [B][Lk] bytes (row stride sm_b), non-zero = ignore, or NULL.  Lq, Lk <= 32, head_dim <= 128.
 * Backward recomputes the probabilities; dq/dk/dv (any may be NULL) are accumulated with the strides given. */
int otter_small_attn_fwd(const float* q, long sq_b, long sq_l, const float* k, const float* v, long sk_b, long sk_l,
                         const unsigned char* key_mask, long sm_b, float* out, long so_b, long so_l, int B, int H, int Lq,
                         int Lk, int head_dim, int causal, void* stream);
int otter_small_attn_bwd(const float* q, long sq_b, long sq_l, const float* k, const float* v, long sk_b, long sk_l,
                         const unsigned char* key_mask, long sm_b, const float* dout, long so_b, long so_l,
                         float* dq, long sdq_b, long sdq_l, float* dk, float* dv, int B, int H, int Lq, int Lk,
                         int head_dim, int causal, void* stream);

/* loss = F.l1_loss / F.mse_loss(pred, gt) (otter.py:378-382), mean over n elements, written to *loss (device);
 * dpred (may be NULL) = grad_scale * d loss / d pred. */
int otter_loss_fwd_bwd(const float* pred, const float* gt, long n, int mse, float grad_scale, float* loss, float* dpred, void* stream);

/* out2[0] = sum of squares of the finite elements of x, out2[1] = number of non-finite elements
 * (get_grad_norm_ and the GradScaler inf check, misc.py:336-343,351-366).  workspace: otter_grad_stats_workspace_floats(). */
long otter_grad_stats_workspace_floats(void);
int otter_grad_stats(const float* x, long n, float* workspace, float* out2, void* stream);

/* torch.optim.AdamW step on flat fp32 buffers (train.py:170-171): grad is multiplied by inv_scale first
 * (GradScaler.unscale_); skip_flag (device float, may be NULL): non-zero skips the update (GradScaler.step). */
int otter_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float inv_scale, const float* skip_flag, void* stream);
int otter_scale_inplace(float* x, long n, float alpha, void* stream);

/* d loss / d temperature of TextAwareVisualExtraction (models.py:17-18,31-32): partial[m*cams + c] = contribution of
 * unit (m, c); operands as otter_text_aware_pool_fwd except pos: fp32 [cams][P][D] (the pos_emb buffers) and
 * grad_out: fp32 [cams][n_obs][K][D]. */
int otter_text_aware_pool_temperature_grad(int dtype, const void* patches, long unit_stride, const float* text, const float* pos,
                                           const float* temperature, const float* grad_out, float* partial, int n_obs, int cams,
                                           int P, int D, int K, int t_rep, void* stream);

/* Centre-crop + bicubic resize of uint8 RGB frames, bit-identical to the host path crop2square + PIL
 * Image.resize((S, S), BICUBIC) of otter_interface.py:98-108 / vision_tf.py:60-67.  src: [n][src_h][src_w][3];
 * crop window (y0, x0, crop_h, crop_w); *_bounds: [S][2] (first source index, count), *_coef: [S][ksize] fixed-point
 * weights (22 fractional bits) as PIL's precompute_coeffs / normalize_coeffs_8bpc produce them (host side:
 * otter_amd.vision_tf.pil_bicubic_coeffs); tmp: [n][crop_h][S][3]; out: [n][3][S][S]. */
int otter_resize_bicubic_u8(const unsigned char* src, int n, int src_h, int src_w, int y0, int x0, int crop_h, int crop_w,
                            const int* h_bounds, const int* h_coef, int h_ksize, const int* v_bounds, const int* v_coef,
                            int v_ksize, unsigned char* tmp, unsigned char* out, int out_size, void* stream);

/* Action post-processing of one control step for n_episodes concurrent episodes, fp64 like the reference's host numpy:
 *   (1) delta chunk -> absolute chunk relative to the current pose: OtterInference._decode_actions
 *       (otter/policy/otter_interface.py:113-133) = convert_abs_action (otter/dataset/utils.py:236-264, with
 *       rot_6d_to_rot_mat / gram_schmidt :19-34,100-119): t = R_cur . dt + t_cur, R = R_cur . R_delta;
 *   (2) which action to execute now: OTTERRolloutPolicy.__call__ (script/droid_inference.py:174-196) -- temporal
 *       ensemble over the last exec_horizon chunks with weights exp(k * i), i = 0 for the oldest (k = 0.05 upstream), or,
 *       receding != 0, the next element of a chunk that is replaced every exec_horizon steps;
 *   (3) action_10d_to_7d (dataset/utils.py:145-151) + `gripper > 0.5` (droid_inference.py:198-199).
 * delta_chunks: fp32 [E][horizon][10] (the policy's output); cur_pose: fp64 [E][10] = xyz, first two rows of the current
 * rotation matrix, gripper (what _proprocess_proprio stores, otter_interface.py:60-84).
 * State, owned by the caller: ring fp64 [E][exec_horizon][exec_horizon][10] and counts int32 [E] (chunks pushed since the
 * episode began: zero an episode's count to restart it; the kernel increments it).
 * Outputs: abs_chunks fp64 [E][horizon][10] (may be NULL), action10 fp64 [E][10] (may be NULL), robot7 fp64 [E][7] =
 * xyz, Euler XYZ, gripper as 0/1. */
int otter_action_step(const float* delta_chunks, const double* cur_pose, double* abs_chunks, double* ring, int* counts,
                      double* action10, double* robot7, int n_episodes, int horizon, int action_dim, int exec_horizon,
                      int receding, double k, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OTTER_HIP_H */
