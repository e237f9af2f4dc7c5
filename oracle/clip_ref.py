"""Functional fp32 restatement of the two CLIP towers as OTTER uses them.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

The OpenAI ``clip`` package is absent from the reference tree (third-party,
``pyproject.toml:21``), so this restates the published architecture and is
anchored on the reference's call sites:

* ``otter/policy/otter.py:141-143``  hook on ``visual.transformer.resblocks[-1].attn``
  -> the LAST block's attention output (after out_proj, before the residual add).
* ``otter/policy/otter.py:144-146``  hook on ``clip_model.transformer`` -> text
  transformer output before ``ln_final``.
* ``otter/policy/otter.py:189-214``  how those activations are post-processed.

All tensors are batch-first here; the reference's (L, N, D) permutes are layout
only.  ``sd`` is a flat ``{key: tensor}`` mapping in the OpenAI key layout,
``prefix`` selects a sub-tree (e.g. ``"clip_model."``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def quick_gelu(x: Tensor) -> Tensor:
    """OpenAI CLIP activation: x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(1.702 * x)


def gelu_erf(x: Tensor) -> Tensor:
    """Exact GELU used by the OpenCLIP datacomp ViT-L/14 weights."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def _count_blocks(sd: Dict[str, Tensor], prefix: str) -> int:
    n = 0
    while f"{prefix}resblocks.{n}.ln_1.weight" in sd:
        n += 1
    return n


def mha_self(x: Tensor, sd: Dict[str, Tensor], p: str, heads: int, causal: bool) -> Tensor:
    """``nn.MultiheadAttention(x, x, x)`` with packed in_proj, batch-first.

    Returns the attention output AFTER out_proj (what a forward hook on the
    attention module sees as ``output[0]``).
    """
    n, l, d = x.shape
    hd = d // heads
    qkv = x @ sd[p + "in_proj_weight"].t() + sd[p + "in_proj_bias"]
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(n, l, heads, hd).transpose(1, 2)
    k = k.view(n, l, heads, hd).transpose(1, 2)
    v = v.view(n, l, heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if causal:
        mask = torch.full((l, l), float("-inf"), dtype=s.dtype).triu(1)
        s = s + mask
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(n, l, d)
    return o @ sd[p + "out_proj.weight"].t() + sd[p + "out_proj.bias"]


def _mlp(x: Tensor, sd: Dict[str, Tensor], p: str, act) -> Tensor:
    h = x @ sd[p + "c_fc.weight"].t() + sd[p + "c_fc.bias"]
    return act(h) @ sd[p + "c_proj.weight"].t() + sd[p + "c_proj.bias"]


def transformer(x: Tensor, sd: Dict[str, Tensor], prefix: str, heads: int, causal: bool,
                act=quick_gelu, return_last_attn: bool = False):
    """Stack of pre-LN residual attention blocks.

    With ``return_last_attn`` the function returns ``(x_out, attn_out_last)``
    where ``attn_out_last`` is the last block's attention-module output.
    """
    nb = _count_blocks(sd, prefix)
    last_attn = None
    for i in range(nb):
        p = f"{prefix}resblocks.{i}."
        a = mha_self(layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]), sd, p + "attn.", heads, causal)
        if i == nb - 1:
            last_attn = a
        x = x + a
        x = x + _mlp(layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]), sd, p + "mlp.", act)
    if return_last_attn:
        return x, last_attn
    return x


def vision_embed(images: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """conv1 (stride = kernel, no bias) + class token + positional embedding + ln_pre."""
    v = prefix + "visual."
    w = sd[v + "conv1.weight"]
    ps = w.shape[-1]
    x = F.conv2d(images, w, bias=None, stride=ps)            # (N, W, g, g)
    x = x.flatten(2).transpose(1, 2)                          # (N, g*g, W)
    cls = sd[v + "class_embedding"].to(x.dtype).expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[v + "positional_embedding"]
    return layer_norm(x, sd[v + "ln_pre.weight"], sd[v + "ln_pre.bias"])


def vision_last_attn(images: Tensor, sd: Dict[str, Tensor], prefix: str = "clip_model.",
                     act=quick_gelu) -> Tensor:
    """(N, 1+P, width): output of the last vision block's attention module.

    This is ``self.activation['image_patches'][0]`` of ``otter.py:203`` (after
    the (L,N,D)->(N,L,D) permute of ``otter.py:204``).
    """
    v = prefix + "visual."
    width = sd[v + "conv1.weight"].shape[0]
    heads = width // 64
    x = vision_embed(images, sd, prefix)
    _, last = transformer(x, sd, v + "transformer.", heads, causal=False, act=act, return_last_attn=True)
    return last


def encode_image(images: Tensor, sd: Dict[str, Tensor], prefix: str = "clip_model.",
                 act=quick_gelu) -> Tensor:
    """Full ``encode_image`` (pooled CLS feature); only used to cross-check the tower."""
    v = prefix + "visual."
    width = sd[v + "conv1.weight"].shape[0]
    heads = width // 64
    x = vision_embed(images, sd, prefix)
    x = transformer(x, sd, v + "transformer.", heads, causal=False, act=act)
    x = layer_norm(x[:, 0], sd[v + "ln_post.weight"], sd[v + "ln_post.bias"])
    return x @ sd[v + "proj"]


def text_transformer_out(tokens: Tensor, sd: Dict[str, Tensor], prefix: str = "clip_model.",
                         act=quick_gelu, heads: Optional[int] = None) -> Tensor:
    """(B, 77, width): text transformer output before ``ln_final`` (hook of ``otter.py:144``)."""
    emb = sd[prefix + "token_embedding.weight"]
    width = emb.shape[1]
    heads = heads or width // 64
    x = emb[tokens.long()] + sd[prefix + "positional_embedding"][: tokens.shape[1]]
    return transformer(x, sd, prefix + "transformer.", heads, causal=True, act=act)


def encode_text(tokens: Tensor, sd: Dict[str, Tensor], prefix: str = "clip_model.",
                act=quick_gelu) -> Tensor:
    """Full ``encode_text`` (EOT-pooled feature); cross-check only."""
    x = text_transformer_out(tokens, sd, prefix, act)
    x = layer_norm(x, sd[prefix + "ln_final.weight"], sd[prefix + "ln_final.bias"])
    eot = tokens.argmax(dim=-1)
    return x[torch.arange(x.shape[0]), eot] @ sd[prefix + "text_projection"]


def patch_features(images: Tensor, sd: Dict[str, Tensor], prefix: str = "clip_model.",
                   act=quick_gelu) -> Tensor:
    """(N, P, embed) L2-normalised patch tokens, ``otter.py:203-214``."""
    v = prefix + "visual."
    x = vision_last_attn(images, sd, prefix, act)[:, 1:]
    x = layer_norm(x, sd[v + "ln_post.weight"], sd[v + "ln_post.bias"])
    if (v + "proj") in sd:
        x = x @ sd[v + "proj"]
    return x / x.norm(dim=-1, keepdim=True)


def text_features(tokens: Tensor, first_k: int, sd: Dict[str, Tensor], prefix: str = "clip_model.",
                  act=quick_gelu) -> Tensor:
    """(B, first_k, embed) L2-normalised per-token text features, ``otter.py:193-195``."""
    x = text_transformer_out(tokens, sd, prefix, act)[:, :first_k]
    x = layer_norm(x, sd[prefix + "ln_final.weight"], sd[prefix + "ln_final.bias"])
    x = x @ sd[prefix + "text_projection"]
    return x / x.norm(dim=-1, keepdim=True)
