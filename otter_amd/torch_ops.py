"""``torch.library`` registration of the hot-path kernels: ``torch.ops.otter_hip.*``.

The same C-ABI entry points as :mod:`otter_amd.ops`, exposed as PyTorch custom operators (functional form: every
operator returns new tensors) with shape-inference ("fake") implementations, so that code written against
``torch.ops`` -- the reference's modules, ``torch.compile`` graphs, exporters -- can call them.  Only a device
implementation is registered: calling an operator on CPU tensors raises (there is no CPU implementation of this path).

    import otter_amd.torch_ops                      # registers the namespace
    y = torch.ops.otter_hip.gemm(a, w_packed, n, bias, 0, False, None)

| operator | reference arithmetic |
|---|---|
| ``gemm(a, w_packed, n, bias, act, out_f32, residual)`` | every ``nn.Linear`` / ``conv1`` of the path, activation / residual fused |
| ``layernorm(x, gamma, beta, eps, out_dtype)`` | ``LayerNorm`` (``out_dtype``: 0 bf16, 1 fp16, 2 fp32) |
| ``attn_fwd_d64(qkv, n_seq, seq_len, heads, causal)`` | CLIP ``nn.MultiheadAttention`` (head width 64) on packed q|k|v rows |
| ``attn_policy(qkv_cache, n_query, pos0, heads)`` | causal attention of the policy transformer over the K/V history |
| ``pos_emb_pack(pos, dtype)`` | layout of ``TextAwareVisualExtraction.pos_emb`` for the pooling kernel |
| ``text_aware_pool(patches, text, pos_packed, temperature, n_obs, cams, t_rep, n_patches)`` | ``TextAwareVisualExtraction.forward`` + the L2 normalisations (``models.py:22-36``, ``otter.py:195,214``) |
| ``l2norm_rows(x)`` | ``F.normalize(x, dim=-1)`` |
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

Tensor = torch.Tensor
_DT = {0: torch.bfloat16, 1: torch.float16, 2: torch.float32}
_lib = torch.library.Library("otter_hip", "DEF")

_lib.define("gemm(Tensor a, Tensor w_packed, int n, Tensor? bias, int act, bool out_f32, Tensor? residual) -> Tensor")
_lib.define("layernorm(Tensor x, Tensor? gamma, Tensor? beta, float eps, int out_dtype) -> Tensor")
_lib.define("attn_fwd_d64(Tensor qkv, int n_seq, int seq_len, int heads, bool causal) -> Tensor")
_lib.define("attn_policy(Tensor qkv_cache, int n_query, int pos0, int heads) -> Tensor")
_lib.define("pos_emb_pack(Tensor pos, int dtype) -> Tensor")
_lib.define("text_aware_pool(Tensor patches, Tensor text, Tensor pos_packed, Tensor temperature, int n_obs, int cams, int t_rep, "
            "int n_patches) -> Tensor")
_lib.define("l2norm_rows(Tensor x) -> Tensor")


def _gemm_out_dtype(a: Tensor, out_f32: bool, residual: Optional[Tensor]):
    if out_f32 or a.dtype == torch.float32 or (residual is not None and residual.dtype == torch.float32):
        return torch.float32
    return a.dtype


# ------------------------------------------------------------------ device implementations
def _gemm(a, w_packed, n, bias, act, out_f32, residual):
    return ops.gemm(a, w_packed, n, bias, act=act, out_f32=out_f32, residual=residual)


def _layernorm(x, gamma, beta, eps, out_dtype):
    return ops.layernorm(x, gamma, beta, eps=eps, out_dtype=_DT[out_dtype])


def _attn_fwd_d64(qkv, n_seq, seq_len, heads, causal):
    return ops.attn_d64(qkv, n_seq, seq_len, heads, causal)


def _attn_policy(qkv_cache, n_query, pos0, heads):
    return ops.attn_policy(qkv_cache, n_query, pos0, heads)


def _pos_emb_pack(pos, dtype):
    return ops.pos_emb_pack(pos, _DT[dtype])


def _text_aware_pool(patches, text, pos_packed, temperature, n_obs, cams, t_rep, n_patches):
    return ops.text_aware_pool(patches, text, pos_packed, temperature, n_obs, cams, t_rep, n_patches=n_patches)


def _l2norm_rows(x):
    return ops.l2norm_rows(x)


for _name, _fn in (("gemm", _gemm), ("layernorm", _layernorm), ("attn_fwd_d64", _attn_fwd_d64), ("attn_policy", _attn_policy),
                   ("pos_emb_pack", _pos_emb_pack), ("text_aware_pool", _text_aware_pool), ("l2norm_rows", _l2norm_rows)):
    _lib.impl(_name, _fn, "CUDA")            # "CUDA" is PyTorch's dispatch key for HIP devices on ROCm builds


# ------------------------------------------------------------------ shape inference (meta / fake tensors)
@torch.library.register_fake("otter_hip::gemm")
def _(a, w_packed, n, bias, act, out_f32, residual):
    return a.new_empty((a.shape[0], n), dtype=_gemm_out_dtype(a, out_f32, residual))


@torch.library.register_fake("otter_hip::layernorm")
def _(x, gamma, beta, eps, out_dtype):
    return x.new_empty(x.shape, dtype=_DT[out_dtype])


@torch.library.register_fake("otter_hip::attn_fwd_d64")
def _(qkv, n_seq, seq_len, heads, causal):
    return qkv.new_empty((qkv.shape[0], heads * 64))


@torch.library.register_fake("otter_hip::attn_policy")
def _(qkv_cache, n_query, pos0, heads):
    b, _, w3 = qkv_cache.shape
    return qkv_cache.new_empty((b * n_query, w3 // 3))


@torch.library.register_fake("otter_hip::pos_emb_pack")
def _(pos, dtype):
    p, d = pos.shape
    return pos.new_empty((-(-p // 16) * 16 * d,), dtype=_DT[dtype])


@torch.library.register_fake("otter_hip::text_aware_pool")
def _(patches, text, pos_packed, temperature, n_obs, cams, t_rep, n_patches):
    return text.new_empty((cams, n_obs, text.shape[1], text.shape[2]), dtype=torch.float32)


@torch.library.register_fake("otter_hip::l2norm_rows")
def _(x):
    return x.new_empty(x.shape, dtype=torch.float32)
