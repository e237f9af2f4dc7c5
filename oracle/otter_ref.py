"""Functional fp32 restatement of OTTER's observation -> action-chunk path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Operates on a flat
``state_dict`` in the reference's checkpoint key layout.  Each function cites
the reference lines it restates (paths under the reference repository).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import clip_ref
from .clip_ref import gelu_erf, layer_norm, quick_gelu

Tensor = torch.Tensor

# otter/policy/vision_tf.py:4,10-11 (values of timm's OPENAI_CLIP_MEAN / _STD)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass
class Dims:
    """The handful of hyper-parameters the arithmetic needs.

    Defaults are those of ``otter/util/args.py:121-178`` (ModelConfig) and
    ``:219-264`` (SharedConfig).
    """
    first_k_tokens: int = 15
    pooling_heads: int = 8
    transformer_heads: int = 8
    action_dim: int = 10
    action_horizon: int = 12
    seq_length: int = 12
    num_cameras: int = 2
    pool_true_text: bool = False
    use_mse: bool = False
    sot_token: int = 49406
    eot_token: int = 49407
    clip_act: str = "quick_gelu"

    @property
    def act(self):
        return quick_gelu if self.clip_act == "quick_gelu" else gelu_erf


# ----------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------

def clip_transform(obs: Tensor) -> Tensor:
    """``otter/policy/vision_tf.py:28-52``: /255 when non-float or max>1, then (x-mean)/std."""
    if not torch.is_floating_point(obs) or obs.max() > 1.0:
        obs = obs.float() / 255.0
    shape = [1] * (obs.dim() - 3) + [3, 1, 1]
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(shape)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(shape)
    return (obs - mean) / std


def create_text_mask(text: Tensor, sot: int, eot: int, first_k: int) -> Tensor:
    """``otter/policy/otter.py:13-38``: True at SOT / EOT / padding(0); first k positions."""
    assert text.dim() in (1, 2)
    m = (text == sot) | (text == eot) | (text == 0)
    return m[..., :first_k]


def sincos_position_embedding(seq_len: int, dim: int) -> Tensor:
    """``otter/policy/models.py:5-12``: [sin | cos] halves (concatenated, not interleaved)."""
    pos = torch.arange(seq_len, dtype=torch.float32)
    inv = 1.0 / (10000 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    ang = pos[:, None] * inv[None, :]
    return torch.cat([ang.sin(), ang.cos()], dim=-1)


def sinusoidal_pe_table(max_len: int, dim: int) -> Tensor:
    """``otter/policy/transformer.py:12-21``: interleaved (even = sin, odd = cos), shape (1, L, D)."""
    position = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2) * (-math.log(10000.0) / dim))
    pe = torch.zeros(max_len, dim)
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe.unsqueeze(0)


def text_aware_visual_extraction(patches: Tensor, text: Tensor, temperature: Tensor, pos_emb: Tensor) -> Tensor:
    """``otter/policy/models.py:22-36``.

    patches (M, P, D) and text (M, K, D) are already L2-normalised by the caller.
    """
    sim = torch.einsum("mkd,mpd->mkp", text, patches)
    attn = torch.softmax(sim / temperature.clamp(0, 100), dim=-1)
    return torch.einsum("mkp,mpd->mkd", attn, patches + pos_emb)


def _cross_mha(q: Tensor, kv: Tensor, sd: Dict[str, Tensor], p: str, heads: int,
               key_padding_mask: Optional[Tensor]) -> Tensor:
    """``nn.MultiheadAttention(d, heads, batch_first=True, kdim=vdim=D_in)`` as built at
    ``otter/policy/transformer.py:165-171`` (separate q/k/v projection weights, packed bias)."""
    b, nq, d = q.shape
    nk = kv.shape[1]
    hd = d // heads
    bias = sd[p + "in_proj_bias"]
    qq = q @ sd[p + "q_proj_weight"].t() + bias[:d]
    kk = kv @ sd[p + "k_proj_weight"].t() + bias[d:2 * d]
    vv = kv @ sd[p + "v_proj_weight"].t() + bias[2 * d:]
    qq = qq.view(b, nq, heads, hd).transpose(1, 2)
    kk = kk.view(b, nk, heads, hd).transpose(1, 2)
    vv = vv.view(b, nk, heads, hd).transpose(1, 2)
    s = (qq @ kk.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = (a @ vv).transpose(1, 2).reshape(b, nq, d)
    return o @ sd[p + "out_proj.weight"].t() + sd[p + "out_proj.bias"]


def attention_pooling(x: Tensor, sd: Dict[str, Tensor], prefix: str, heads: int,
                      mask: Optional[Tensor] = None) -> Tensor:
    """``otter/policy/transformer.py:162-232`` (CrossAttentionBlock x N + final LN + flatten).

    Note the residual adds onto the *normalised* query (``transformer.py:189-192``).
    """
    b = x.shape[0]
    q = sd[prefix + "query"].expand(b, -1, -1)
    i = 0
    while f"{prefix}blocks.{i}.q_layer_norm.weight" in sd:
        p = f"{prefix}blocks.{i}."
        q = layer_norm(q, sd[p + "q_layer_norm.weight"], sd[p + "q_layer_norm.bias"])
        kv = layer_norm(x, sd[p + "kv_layer_norm.weight"], sd[p + "kv_layer_norm.bias"])
        q = q + _cross_mha(q, kv, sd, p + "attention.", heads, mask)
        h = layer_norm(q, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])
        h = gelu_erf(h @ sd[p + "mlp.fc1.weight"].t() + sd[p + "mlp.fc1.bias"])
        h = h @ sd[p + "mlp.fc2.weight"].t() + sd[p + "mlp.fc2.bias"]
        q = q + h
        i += 1
    q = layer_norm(q, sd[prefix + "layer_norm.weight"], sd[prefix + "layer_norm.bias"])
    return q.reshape(b, -1)


def proprio_encoder(x: Tensor, sd: Dict[str, Tensor], prefix: str = "proprio_encoder.") -> Tensor:
    """``otter/policy/models.py:38-51``: Linear-ReLU-Linear-ReLU-Linear."""
    h = torch.relu(x @ sd[prefix + "mlp.0.weight"].t() + sd[prefix + "mlp.0.bias"])
    h = torch.relu(h @ sd[prefix + "mlp.2.weight"].t() + sd[prefix + "mlp.2.bias"])
    return h @ sd[prefix + "mlp.4.weight"].t() + sd[prefix + "mlp.4.bias"]


def action_head(x: Tensor, sd: Dict[str, Tensor], prefix: str = "action_head.") -> Tensor:
    """``otter/policy/models.py:53-67`` without the final reshape (callers view it)."""
    h = torch.relu(x @ sd[prefix + "mlp.0.weight"].t() + sd[prefix + "mlp.0.bias"])
    return h @ sd[prefix + "mlp.2.weight"].t() + sd[prefix + "mlp.2.bias"]


def causal_transformer(x: Tensor, sd: Dict[str, Tensor], heads: int, prefix: str = "policy.") -> Tensor:
    """``otter/policy/transformer.py:53-160``: pre-LN blocks, causal SDPA, GELU(erf) MLP."""
    b, t, d = x.shape
    hd = d // heads
    i = 0
    while f"{prefix}layers.{i}.pre_ln.weight" in sd:
        p = f"{prefix}layers.{i}."
        h = layer_norm(x, sd[p + "pre_ln.weight"], sd[p + "pre_ln.bias"])
        q = (h @ sd[p + "attention.query.weight"].t() + sd[p + "attention.query.bias"]).view(b, t, heads, hd).transpose(1, 2)
        k = (h @ sd[p + "attention.key.weight"].t() + sd[p + "attention.key.bias"]).view(b, t, heads, hd).transpose(1, 2)
        v = (h @ sd[p + "attention.value.weight"].t() + sd[p + "attention.value.bias"]).view(b, t, heads, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
        s = s + torch.full((t, t), float("-inf")).triu(1)
        o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(b, t, d)
        x = x + (o @ sd[p + "attention.out.weight"].t() + sd[p + "attention.out.bias"])
        h = layer_norm(x, sd[p + "post_ln.weight"], sd[p + "post_ln.bias"])
        h = gelu_erf(h @ sd[p + "mlp.0.weight"].t() + sd[p + "mlp.0.bias"])
        x = x + (h @ sd[p + "mlp.2.weight"].t() + sd[p + "mlp.2.bias"])
        i += 1
    return x


# ----------------------------------------------------------------------------
# the path
# ----------------------------------------------------------------------------

def extract_clip_features(images: Tensor, text: Tensor, sd: Dict[str, Tensor], dims: Dims
                          ) -> Tuple[List[Tensor], Tensor]:
    """``otter/policy/otter.py:177-222``.

    images (B, T, cams, 3, H, W) float (already normalised), text (B, 77) int.
    Returns ([cams x (B*T, P, D)], (B*T, K, D)).
    """
    b, t, cams = images.shape[:3]
    flat = images.reshape(b * t * cams, *images.shape[3:])
    tf = clip_ref.text_features(text, dims.first_k_tokens, sd, "clip_model.", dims.act)
    tf = tf.repeat_interleave(t, dim=0)
    pf = clip_ref.patch_features(flat, sd, "clip_model.", dims.act)
    pf = pf.view(b * t, cams, *pf.shape[1:])
    return [pf[:, i] for i in range(cams)], tf


def forward_encoder(images: Tensor, text: Tensor, proprio: Tensor, sd: Dict[str, Tensor], dims: Dims,
                    text_mask: Optional[Tensor] = None, clip_features: Optional[Tuple[List[Tensor], Tensor]] = None) -> Tensor:
    """``otter/policy/otter.py:224-267`` -> transformer input (B, T, transformer_dim).

    ``clip_features`` (what :func:`extract_clip_features` returns) replaces the frozen towers' output: a test can hand in
    the features another implementation produced and compare only what comes after the towers."""
    b, t = images.shape[:2]
    per_cam, tf = clip_features if clip_features is not None else extract_clip_features(images, text, sd, dims)
    if dims.pool_true_text:
        text_mask = text_mask.repeat_interleave(t, dim=0)
    else:
        text_mask = None if text_mask is None else text_mask
    vis = []
    for c, feats in enumerate(per_cam):
        ta = text_aware_visual_extraction(
            feats, tf, sd[f"visual_extraction.{c}.temperature"], sd[f"visual_extraction.{c}.pos_emb"])
        vis.append(attention_pooling(ta, sd, f"vision_poolings.{c}.", dims.pooling_heads, text_mask))
    vision_token = torch.cat(vis, dim=-1)
    text_token = attention_pooling(tf, sd, "text_pooling.", dims.pooling_heads, text_mask)
    proprio_token = proprio_encoder(proprio.reshape(b * t, -1), sd)
    combined = torch.cat([text_token, vision_token, proprio_token], dim=-1)
    x = combined @ sd["input_projection.weight"].t() + sd["input_projection.bias"]
    return x.view(b, t, -1)


def forward_actions(images: Tensor, text: Tensor, proprio: Tensor, sd: Dict[str, Tensor], dims: Dims,
                    text_mask: Optional[Tensor] = None, clip_features=None) -> Tensor:
    """``otter/policy/otter.py:269-291`` -> (B, T, horizon, action_dim)."""
    b, t = images.shape[:2]
    x = forward_encoder(images, text, proprio, sd, dims, text_mask, clip_features)
    x = x + sd["pe.pe"][:, :t]
    x = causal_transformer(x, sd, dims.transformer_heads)
    a = action_head(x, sd)
    return a.view(b, t, dims.action_horizon, dims.action_dim)


def forward(images, text: Tensor, proprio: Tensor, sd: Dict[str, Tensor], dims: Dims,
            gt_actions: Optional[Tensor] = None, camera_keys: Optional[Sequence[str]] = None, clip_features=None) -> Tensor:
    """``otter/policy/otter.py:339-383`` with ``text`` already tokenised."""
    if isinstance(images, dict):
        keys = sorted(camera_keys if camera_keys is not None else images.keys())
        images = torch.stack([images[k] for k in keys], dim=2)
        if images.shape[-1] == 3:
            images = images.permute(0, 1, 2, 5, 3, 4)
    mask = create_text_mask(text, dims.sot_token, dims.eot_token, dims.first_k_tokens) if dims.pool_true_text else None
    if not torch.is_floating_point(images):
        images = clip_transform(images)
    actions = forward_actions(images, text, proprio, sd, dims, mask, clip_features)
    if gt_actions is None:
        return actions
    if dims.use_mse:
        return torch.nn.functional.mse_loss(actions, gt_actions)
    return torch.nn.functional.l1_loss(actions, gt_actions)


class Rollout:
    """Single-episode stateful inference, ``otter/policy/otter.py:153-175`` and ``:293-337``.

    Keeps the cache of transformer *inputs* (pre positional-embedding), truncated to the
    last ``seq_length`` steps; every call re-encodes the whole window.
    """

    def __init__(self, sd: Dict[str, Tensor], dims: Dims):
        self.sd, self.dims = sd, dims
        self.cached: Optional[Tensor] = None
        self.cache_size = 0

    def reset(self):
        self.cached, self.cache_size = None, 0

    @torch.no_grad()
    def step(self, images: Tensor, text: Tensor, proprio: Tensor, text_mask: Optional[Tensor] = None) -> Tensor:
        d = self.dims
        x = forward_encoder(images[None, None], text[None], proprio[None, None], self.sd, d,
                            None if text_mask is None else text_mask[None])
        self.cached = x if self.cached is None else torch.cat([self.cached, x], dim=1)[:, -d.seq_length:]
        self.cache_size = min(self.cache_size + 1, d.seq_length)
        h = self.cached + self.sd["pe.pe"][:, : self.cached.shape[1]]
        h = causal_transformer(h, self.sd, d.transformer_heads)
        a = action_head(h[:, self.cache_size - 1], self.sd)
        return a.reshape(d.action_horizon, d.action_dim)
