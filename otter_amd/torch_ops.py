"""``torch.library`` registration of the hot-path kernels: ``torch.ops.otter_hip.*``.

Every C-ABI entry point the observation -> action path launches is a PyTorch custom operator here, with a
shape-inference ("fake") implementation, so that the model code (:mod:`otter_amd.device_model`, the rollout engine) is
ordinary torch-op code: ``torch.compile`` / export see the kernels as graph nodes, and code written against ``torch.ops``
can call them.  Only a device implementation is registered: calling an operator on CPU tensors raises (there is no CPU
implementation of this path).  Underneath, each operator is the tensor -> pointer wrapper of :mod:`otter_amd.ops` (ctypes
into ``libotter_hip.so``).

Operators that produce a tensor return it; operators that write into caller-owned storage (the in-place residual stream, a
row of the K/V cache, a slice of the concatenated feature buffer, the rollout windows) declare that argument mutable
(``Tensor(a!)``) and return nothing.  ``out_dtype`` / ``dtype`` arguments use the C ABI's codes: 0 bf16, 1 fp16, 2 fp32.

| operator | reference arithmetic |
|---|---|
| ``gemm``, ``gemm_into``, ``gemm_residual_``, ``gemm_residual_stats_`` | every ``nn.Linear`` / ``conv1`` of the path; bias, activation, residual add and a LayerNorm folded into the weights are epilogues |
| ``row_stats`` | mean / rstd of ``ln_1`` / ``ln_2`` rows for the folded LayerNorm |
| ``layernorm`` | ``LayerNorm`` |
| ``im2col_patches`` | ``clip_transform`` + patch extraction for ``visual.conv1`` (``vision_tf.py:28-52``) |
| ``vit_embed_lnpre`` | class token + positional embedding + ``ln_pre`` |
| ``text_embed`` | ``token_embedding(ids) + positional_embedding`` |
| ``attn_fwd_d64`` | CLIP ``nn.MultiheadAttention`` (head width 64) on packed q|k|v rows |
| ``pos_emb_pack``, ``text_aware_pool`` | ``TextAwareVisualExtraction`` + the L2 normalisations (``models.py:22-36``, ``otter.py:195,214``) |
| ``l2norm_rows`` | ``F.normalize(x, dim=-1)`` |
| ``readout_pool_multi`` | query side of every ``AttentionPooling`` of one encoder call (``transformer.py:162-232``) |
| ``linear_f32``, ``linear_f32_into`` | ``ProprioceptionEncoder`` layers (``models.py:38-51``) |
| ``repeat_rows_into``, ``convert_rows`` | ``repeat_interleave`` of the pooled text token; row selection in front of the action head |
| ``add_pe``, ``add_pe_rows`` | ``SinusoidalPositionalEmbedding`` on the window (``transformer.py:23-24``) |
| ``attn_policy``, ``attn_policy_rows`` | causal attention of the policy transformer over the GPU-resident K/V history |
| ``window_append`` | ``OTTER.update_cache`` (``otter.py:158-175``) |
| ``action_step`` | ``convert_abs_action`` + the rollout script's action queue + 10-d -> 7-d (``droid_inference.py:174-200``) |
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops

Tensor = torch.Tensor
_DT = {0: torch.bfloat16, 1: torch.float16, 2: torch.float32}
_CODE = {v: k for k, v in _DT.items()}
_lib = torch.library.Library("otter_hip", "DEF")

_SCHEMAS = (
    "gemm(Tensor a, Tensor w_packed, int n, Tensor? bias, int act, bool out_f32, Tensor? residual, Tensor? ln_stats=None, Tensor? ln_colsum=None) -> Tensor",
    "gemm_into(Tensor(a!) out, Tensor a, Tensor w_packed, int n, Tensor? bias, int act, Tensor? residual, Tensor? ln_stats=None, Tensor? ln_colsum=None) -> ()",
    "gemm_residual_(Tensor(a!) x, Tensor a, Tensor w_packed, int n, Tensor? bias) -> ()",
    "gemm_residual_stats_(Tensor(a!) x, Tensor a, Tensor w_packed, int n, Tensor? bias, float eps) -> Tensor",
    "row_stats(Tensor x, float eps) -> Tensor",
    "layernorm(Tensor x, Tensor? gamma, Tensor? beta, float eps, int out_dtype) -> Tensor",
    "im2col_patches(Tensor images, int patch, int kpad, float[] mean, float[] std, int out_dtype) -> Tensor",
    "vit_embed_lnpre(Tensor patch_tok, Tensor cls, Tensor pos, Tensor gamma, Tensor beta, int n_img, int n_patches, float eps, int out_dtype) -> Tensor",
    "text_embed(Tensor ids, int take, Tensor token_embedding, Tensor pos, int out_dtype) -> Tensor",
    "attn_fwd_d64(Tensor qkv, int n_seq, int seq_len, int heads, bool causal) -> Tensor",
    "attn_policy(Tensor qkv_cache, int n_query, int pos0, int heads) -> Tensor",
    "attn_policy_rows(Tensor fresh, Tensor(a!) qkv_cache, Tensor idx, Tensor lens, int heads) -> Tensor",
    "pos_emb_pack(Tensor pos, int dtype) -> Tensor",
    "text_aware_pool(Tensor patches, Tensor text, Tensor pos_packed, Tensor temperature, int n_obs, int cams, int t_rep, int n_patches) -> Tensor",
    "l2norm_rows(Tensor x) -> Tensor",
    "readout_pool_multi(Tensor[] block_params, int[] n_blocks, Tensor[] query, Tensor[] final_w, Tensor[] final_b, Tensor[] kv, Tensor?[] mask, "
    "Tensor(a!)[] out, int[] n_obs, int[] nk, int[] readouts, int[] d, int[] heads, int[] hidden) -> ()",
    "linear_f32(Tensor x, Tensor wt, Tensor? bias, int act) -> Tensor",
    "linear_f32_into(Tensor(a!) out, Tensor x, Tensor wt, Tensor? bias, int act) -> ()",
    "repeat_rows_into(Tensor(a!) dst, Tensor src, int rep) -> ()",
    "convert_rows(Tensor src, int out_dtype, int cols_pad, int rep, Tensor? pick, int group, int rows, int pick_bias) -> Tensor",
    "add_pe(Tensor window, Tensor pe, int n_query, int pos0) -> Tensor",
    "add_pe_rows(Tensor window, Tensor pe, Tensor idx, Tensor lens) -> Tensor",
    "window_append(Tensor src, Tensor fresh, Tensor(a!) dst, Tensor lens_in, Tensor(b!) lens_out) -> ()",
    "action_step(Tensor delta_chunks, Tensor cur_pose, Tensor(a!) ring, Tensor(b!) counts, int exec_horizon, bool receding, float k, bool want_abs) "
    "-> (Tensor, Tensor, Tensor)",
)
for _s in _SCHEMAS:
    _lib.define(_s)
OP_NAMES = tuple(s.split("(", 1)[0] for s in _SCHEMAS)

READOUT_FIELDS = ops.READOUT_FIELDS


def _gemm_out_dtype(a: Tensor, out_f32: bool, residual: Optional[Tensor]):
    if out_f32 or a.dtype == torch.float32 or (residual is not None and residual.dtype == torch.float32):
        return torch.float32
    return a.dtype


# ------------------------------------------------------------------ device implementations
def _gemm(a, w_packed, n, bias, act, out_f32, residual, ln_stats=None, ln_colsum=None):
    return ops.gemm(a, w_packed, n, bias, act=act, out_f32=out_f32, residual=residual, ln_stats=ln_stats, ln_colsum=ln_colsum)


def _gemm_into(out, a, w_packed, n, bias, act, residual, ln_stats=None, ln_colsum=None):
    ops.gemm(a, w_packed, n, bias, act=act, out_f32=out.dtype == torch.float32, residual=residual, out=out, ln_stats=ln_stats, ln_colsum=ln_colsum)


def _gemm_residual_(x, a, w_packed, n, bias):
    ops.gemm(a, w_packed, n, bias, residual=x, out=x)


def _gemm_residual_stats_(x, a, w_packed, n, bias, eps):
    return ops.gemm_residual_stats_(a, w_packed, n, bias, x, eps)


def _row_stats(x, eps):
    return ops.row_stats(x, eps)


def _layernorm(x, gamma, beta, eps, out_dtype):
    return ops.layernorm(x, gamma, beta, eps=eps, out_dtype=_DT[out_dtype])


def _im2col_patches(images, patch, kpad, mean, std, out_dtype):
    return ops.im2col_patches(images, patch, kpad, mean, std, _DT[out_dtype])


def _vit_embed_lnpre(patch_tok, cls, pos, gamma, beta, n_img, n_patches, eps, out_dtype):
    return ops.vit_embed_lnpre(patch_tok, cls, pos, gamma, beta, n_img, n_patches, eps, _DT[out_dtype])


def _text_embed(ids, take, token_embedding, pos, out_dtype):
    return ops.text_embed(ids, take, token_embedding, pos, _DT[out_dtype])


def _attn_fwd_d64(qkv, n_seq, seq_len, heads, causal):
    return ops.attn_d64(qkv, n_seq, seq_len, heads, causal)


def _attn_policy(qkv_cache, n_query, pos0, heads):
    return ops.attn_policy(qkv_cache, n_query, pos0, heads)


def _attn_policy_rows(fresh, qkv_cache, idx, lens, heads):
    return ops.attn_policy_rows(fresh, qkv_cache, idx, lens, heads)


def _pos_emb_pack(pos, dtype):
    return ops.pos_emb_pack(pos, _DT[dtype])


def _text_aware_pool(patches, text, pos_packed, temperature, n_obs, cams, t_rep, n_patches):
    return ops.text_aware_pool(patches, text, pos_packed, temperature, n_obs, cams, t_rep, n_patches=n_patches)


def _l2norm_rows(x):
    return ops.l2norm_rows(x)


def _readout_pool_multi(block_params, n_blocks, query, final_w, final_b, kv, mask, out, n_obs, nk, readouts, d, heads, hidden):
    problems, off = [], 0
    nf = len(READOUT_FIELDS)
    for i in range(len(kv)):
        cnt = n_blocks[i] * nf
        blocks = list(block_params[off:off + cnt])
        off += cnt
        problems.append((blocks, query[i], final_w[i], final_b[i], kv[i], mask[i], out[i], n_obs[i], nk[i], readouts[i], d[i], heads[i], hidden[i]))
    for i in range(0, len(problems), 4):
        ops.readout_pool_multi(problems[i:i + 4])


def _linear_f32(x, wt, bias, act):
    return ops.linear_f32(x, wt, bias, act)


def _linear_f32_into(out, x, wt, bias, act):
    ops.linear_f32(x, wt, bias, act, out=out)


def _repeat_rows_into(dst, src, rep):
    ops.repeat_rows(src, dst, rep)


def _convert_rows(src, out_dtype, cols_pad, rep, pick, group, rows, pick_bias):
    return ops.convert_rows(src, _DT[out_dtype], cols_pad or None, rep, pick, group, rows if rows >= 0 else None, pick_bias)


def _add_pe(window, pe, n_query, pos0):
    return ops.add_pe(window, pe, n_query, pos0)


def _add_pe_rows(window, pe, idx, lens):
    return ops.add_pe_rows(window, pe, idx, lens)


def _window_append(src, fresh, dst, lens_in, lens_out):
    ops.window_append(src, fresh, dst, lens_in, lens_out)


def _action_step(delta_chunks, cur_pose, ring, counts, exec_horizon, receding, k, want_abs):
    robot, act10, abs_chunks = ops.action_step(delta_chunks, cur_pose, ring, counts, exec_horizon, receding, k, want_abs)
    if abs_chunks is None:
        abs_chunks = delta_chunks.new_empty((0,), dtype=torch.float64)
    return robot, act10, abs_chunks


for _name in OP_NAMES:
    _lib.impl(_name, globals()["_" + _name], "CUDA")            # "CUDA" is PyTorch's dispatch key for HIP devices on ROCm builds


# ------------------------------------------------------------------ shape inference (meta / fake tensors)
def _fake(name):
    return torch.library.register_fake("otter_hip::" + name)


@_fake("gemm")
def _(a, w_packed, n, bias, act, out_f32, residual, ln_stats=None, ln_colsum=None):
    return a.new_empty((a.shape[0], n), dtype=_gemm_out_dtype(a, out_f32, residual))


@_fake("gemm_into")
def _(out, a, w_packed, n, bias, act, residual, ln_stats=None, ln_colsum=None):
    return None


@_fake("gemm_residual_")
def _(x, a, w_packed, n, bias):
    return None


@_fake("gemm_residual_stats_")
def _(x, a, w_packed, n, bias, eps):
    return x.new_empty((x.shape[0], 2), dtype=torch.float32)


@_fake("row_stats")
def _(x, eps):
    return x.new_empty((x.shape[0], 2), dtype=torch.float32)


@_fake("layernorm")
def _(x, gamma, beta, eps, out_dtype):
    return x.new_empty(x.shape, dtype=_DT[out_dtype])


@_fake("im2col_patches")
def _(images, patch, kpad, mean, std, out_dtype):
    n, _, res, _ = images.shape
    g = res // patch
    return images.new_empty((n * g * g, kpad), dtype=_DT[out_dtype])


@_fake("vit_embed_lnpre")
def _(patch_tok, cls, pos, gamma, beta, n_img, n_patches, eps, out_dtype):
    return patch_tok.new_empty((n_img * (n_patches + 1), patch_tok.shape[1]), dtype=_DT[out_dtype])


@_fake("text_embed")
def _(ids, take, token_embedding, pos, out_dtype):
    return token_embedding.new_empty((ids.shape[0] * take, token_embedding.shape[1]), dtype=_DT[out_dtype])


@_fake("attn_fwd_d64")
def _(qkv, n_seq, seq_len, heads, causal):
    return qkv.new_empty((qkv.shape[0], heads * 64))


@_fake("attn_policy")
def _(qkv_cache, n_query, pos0, heads):
    b, _, w3 = qkv_cache.shape
    return qkv_cache.new_empty((b * n_query, w3 // 3))


@_fake("attn_policy_rows")
def _(fresh, qkv_cache, idx, lens, heads):
    return fresh.new_empty((fresh.shape[0], fresh.shape[1] // 3))


@_fake("pos_emb_pack")
def _(pos, dtype):
    p, d = pos.shape
    return pos.new_empty((-(-p // 16) * 16 * d,), dtype=_DT[dtype])


@_fake("text_aware_pool")
def _(patches, text, pos_packed, temperature, n_obs, cams, t_rep, n_patches):
    return text.new_empty((cams, n_obs, text.shape[1], text.shape[2]), dtype=torch.float32)


@_fake("l2norm_rows")
def _(x):
    return x.new_empty(x.shape, dtype=torch.float32)


@_fake("readout_pool_multi")
def _(block_params, n_blocks, query, final_w, final_b, kv, mask, out, n_obs, nk, readouts, d, heads, hidden):
    return None


@_fake("linear_f32")
def _(x, wt, bias, act):
    return x.new_empty((x.shape[0], wt.shape[1]))


@_fake("linear_f32_into")
def _(out, x, wt, bias, act):
    return None


@_fake("repeat_rows_into")
def _(dst, src, rep):
    return None


@_fake("convert_rows")
def _(src, out_dtype, cols_pad, rep, pick, group, rows, pick_bias):
    n_rows = pick.shape[0] if pick is not None else (rows if rows >= 0 else src.shape[0] * rep)
    return src.new_empty((n_rows, cols_pad or src.shape[1]), dtype=_DT[out_dtype])


@_fake("add_pe")
def _(window, pe, n_query, pos0):
    return window.new_empty((window.shape[0] * n_query, window.shape[2]))


@_fake("add_pe_rows")
def _(window, pe, idx, lens):
    return window.new_empty((idx.shape[0], window.shape[2]))


@_fake("window_append")
def _(src, fresh, dst, lens_in, lens_out):
    return None


@_fake("action_step")
def _(delta_chunks, cur_pose, ring, counts, exec_horizon, receding, k, want_abs):
    e, h, a = delta_chunks.shape
    f64 = torch.float64
    return (delta_chunks.new_empty((e, 7), dtype=f64), delta_chunks.new_empty((e, a), dtype=f64),
            delta_chunks.new_empty((e, h, a) if want_abs else (0,), dtype=f64))


# ------------------------------------------------------------------ the call surface the model code uses
class _Namespace:
    """``otter_amd.ops``'s signatures over ``torch.ops.otter_hip.*`` (the ``.default`` overloads: no overload resolution per
    call).  :mod:`otter_amd.device_model` and the rollout engine import this object under the name ``ops``: every kernel
    they launch is a custom-operator call.  Host-side helpers (weight packing at load time, workspaces, constants) pass
    through to :mod:`otter_amd.ops` unchanged."""

    def __init__(self):
        ns = torch.ops.otter_hip
        self._op = {n: getattr(ns, n).default for n in OP_NAMES}

    def __getattr__(self, name):                 # helpers and constants: pack_weight, fold_layernorm, ACT_*, OtterHipError, ...
        return getattr(ops, name)

    def gemm(self, a, w_packed, n, bias=None, *, act=ops.ACT_NONE, out_f32=False, residual=None, out=None, ln_stats=None, ln_colsum=None,
             force_tile_128=False, split_k=True):
        if force_tile_128 or not split_k:        # tuning knobs of tests and experiments: not part of the operator surface
            return ops.gemm(a, w_packed, n, bias, act=act, out_f32=out_f32, residual=residual, out=out, ln_stats=ln_stats, ln_colsum=ln_colsum,
                            force_tile_128=force_tile_128, split_k=split_k)
        if out is None:
            return self._op["gemm"](a, w_packed, n, bias, act, out_f32, residual, ln_stats, ln_colsum)
        if out is residual and act == ops.ACT_NONE and ln_stats is None:
            self._op["gemm_residual_"](out, a, w_packed, n, bias)
        else:
            self._op["gemm_into"](out, a, w_packed, n, bias, act, residual, ln_stats, ln_colsum)
        return out

    def gemm_residual_stats_(self, a, w_packed, n, bias, x, eps=1e-5):
        return self._op["gemm_residual_stats_"](x, a, w_packed, n, bias, eps)

    def row_stats(self, x, eps=1e-5):
        return self._op["row_stats"](x, eps)

    def layernorm(self, x, gamma, beta, *, eps=1e-5, out_dtype=torch.bfloat16, out=None):
        if out is not None:
            return ops.layernorm(x, gamma, beta, eps=eps, out_dtype=out_dtype, out=out)
        return self._op["layernorm"](x, gamma, beta, eps, _CODE[out_dtype])

    def im2col_patches(self, images, patch, kpad, mean, std, out_dtype=torch.bfloat16):
        return self._op["im2col_patches"](images, patch, kpad, [float(v) for v in mean], [float(v) for v in std], _CODE[out_dtype])

    def vit_embed_lnpre(self, patch_tok, cls, pos, gamma, beta, n_img, n_patches, eps=1e-5, out_dtype=torch.float32):
        return self._op["vit_embed_lnpre"](patch_tok, cls, pos, gamma, beta, n_img, n_patches, eps, _CODE[out_dtype])

    def text_embed(self, ids, take, token_embedding, pos, out_dtype=torch.float32):
        return self._op["text_embed"](ids, take, token_embedding, pos, _CODE[out_dtype])

    def attn_d64(self, qkv, n_seq, seq_len, heads, causal):
        return self._op["attn_fwd_d64"](qkv, n_seq, seq_len, heads, bool(causal))

    def attn_policy(self, qkv_cache, n_query, pos0, heads):
        return self._op["attn_policy"](qkv_cache, n_query, pos0, heads)

    def attn_policy_rows(self, fresh, qkv_cache, idx, lens, heads):
        return self._op["attn_policy_rows"](fresh, qkv_cache, idx, lens, heads)

    def pos_emb_pack(self, pos, dtype=torch.bfloat16):
        return self._op["pos_emb_pack"](pos, _CODE[dtype])

    def text_aware_pool(self, patches, text, pos_packed, temperature, n_obs, cams, t_rep, n_patches=None):
        return self._op["text_aware_pool"](patches, text, pos_packed, temperature, n_obs, cams, t_rep,
                                           patches.shape[1] if n_patches is None else n_patches)

    def l2norm_rows(self, x):
        return self._op["l2norm_rows"](x)

    def readout_pool_multi(self, problems) -> None:
        """``problems``: tuples (block_params, query, final_w, final_b, kv, mask, out, n_obs, nk, readouts, d, heads, hidden),
        ``block_params`` the blocks' 12 tensors each in ``READOUT_FIELDS`` order (what ``PoolingWeights.problem`` returns)."""
        nf = len(READOUT_FIELDS)
        cols: List[list] = [[] for _ in range(14)]
        for pr in problems:
            cols[0].extend(pr[0])
            cols[1].append(len(pr[0]) // nf)
            for j in range(1, 13):
                cols[j + 1].append(pr[j])
        self._op["readout_pool_multi"](*cols)

    def linear_f32(self, x, wt, bias, act=ops.ACT_NONE, out=None):
        if out is None:
            return self._op["linear_f32"](x, wt, bias, act)
        self._op["linear_f32_into"](out, x, wt, bias, act)
        return out

    def repeat_rows(self, src, dst, rep):
        self._op["repeat_rows_into"](dst, src, rep)
        return dst

    def convert_rows(self, src, out_dtype=torch.float32, cols_pad=None, rep=1, pick=None, group=1, rows=None, pick_bias=0):
        return self._op["convert_rows"](src, _CODE[out_dtype], cols_pad or 0, rep, pick, group, -1 if rows is None else rows, pick_bias)

    def add_pe(self, window, pe, n_query, pos0):
        return self._op["add_pe"](window, pe, n_query, pos0)

    def add_pe_rows(self, window, pe, idx, lens):
        return self._op["add_pe_rows"](window, pe, idx, lens)

    def window_append(self, src, fresh, dst, lens_in, lens_out) -> None:
        self._op["window_append"](src, fresh, dst, lens_in, lens_out)

    def action_step(self, delta_chunks, cur_pose, ring, counts, exec_horizon, receding, k=0.05, want_abs=False):
        robot, act10, abs_chunks = self._op["action_step"](delta_chunks, cur_pose, ring, counts, int(exec_horizon), bool(receding), float(k), bool(want_abs))
        return robot, act10, (abs_chunks if want_abs else None)


kernels = _Namespace()
