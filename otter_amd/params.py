"""Parameter inventory of an OTTER checkpoint and a module tree that holds it.

The reference saves ``model.state_dict()`` (``otter/util/engine.py:112-122``) and
loads it strictly (``otter/util/misc.py:28-50``), so the key names and shapes are a
file-format contract.  This module enumerates them from the configuration and
builds a bare ``nn.Module`` tree with exactly those names; the compute lives in the
HIP kernels, not in the tree.

Key layout (sources: ``otter/policy/otter.py:72-129``, ``otter/policy/models.py``,
``otter/policy/transformer.py`` and the OpenAI CLIP module tree they instantiate):

* ``clip_model.*``                 frozen CLIP towers (fp32)
* ``text_pooling.*``               AttentionPooling over text tokens
* ``visual_extraction.{c}.*``      temperature + ``pos_emb`` buffer per camera
* ``vision_poolings.{c}.*``        AttentionPooling per camera
* ``proprio_encoder.mlp.{0,2,4}.*``, ``input_projection.*``, ``pe.pe``
* ``policy.layers.{i}.*``          causal transformer
* ``action_head.mlp.{0,2}.*``
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Iterator, List, Optional, Tuple

import torch
import torch.nn as nn

from .config import ClipArch, ModelConfig, SharedConfig, clip_arch

Tensor = torch.Tensor
Init = Callable[[Tuple[int, ...], torch.Generator], Tensor]


@dataclass
class ParamSpec:
    key: str
    shape: Tuple[int, ...]
    init: Init
    kind: str = "param"        # "param" (trainable), "frozen" (requires_grad False) or "buffer"


# ---------------------------------------------------------------------------- init recipes

def _normal(std: float) -> Init:
    return lambda shape, g: torch.randn(shape, generator=g) * std


def _zeros(shape, g):
    return torch.zeros(shape)


def _ones(shape, g):
    return torch.ones(shape)


def _const(v: float) -> Init:
    return lambda shape, g: torch.full(shape, float(v))


def _uniform(bound: float) -> Init:
    return lambda shape, g: (torch.rand(shape, generator=g) * 2 - 1) * bound


def _linear_weight(fan_in: int) -> Init:
    # torch.nn.Linear default: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    return _uniform(1.0 / math.sqrt(fan_in))


def _xavier(fan_in: int, fan_out: int) -> Init:
    return _uniform(math.sqrt(6.0 / (fan_in + fan_out)))


def _fixed(t: Tensor) -> Init:
    return lambda shape, g: t.clone()


def _linear(prefix: str, out_f: int, in_f: int, kind="param") -> List[ParamSpec]:
    return [ParamSpec(prefix + "weight", (out_f, in_f), _linear_weight(in_f), kind),
            ParamSpec(prefix + "bias", (out_f,), _linear_weight(in_f), kind)]


def _ln(prefix: str, d: int, kind="param") -> List[ParamSpec]:
    return [ParamSpec(prefix + "weight", (d,), _ones, kind), ParamSpec(prefix + "bias", (d,), _zeros, kind)]


# ---------------------------------------------------------------------------- tables

def sincos_position_embedding(seq_len: int, dim: int) -> Tensor:
    """[sin | cos] table of the text-aware extraction (``otter/policy/models.py:5-12``)."""
    pos = torch.arange(seq_len, dtype=torch.float32)
    inv = 1.0 / (10000 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    ang = torch.outer(pos, inv)
    return torch.cat([ang.sin(), ang.cos()], dim=-1)


def sinusoidal_table(max_len: int, dim: int) -> Tensor:
    """Interleaved sin/cos table of the policy input (``otter/policy/transformer.py:12-21``)."""
    position = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2) * (-math.log(10000.0) / dim))
    pe = torch.zeros(max_len, dim)
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe.unsqueeze(0)


# ---------------------------------------------------------------------------- inventories

def clip_specs(a: ClipArch, prefix: str = "clip_model.") -> List[ParamSpec]:
    """Every tensor of an OpenAI-layout CLIP ``state_dict`` (all frozen, ``otter.py:77-80``)."""
    F = "frozen"
    out: List[ParamSpec] = []
    vw, tw = a.vision_width, a.transformer_width
    ntok = a.grid * a.grid + 1
    s = vw ** -0.5
    out += [
        ParamSpec(prefix + "positional_embedding", (a.context_length, tw), _normal(0.01), F),
        ParamSpec(prefix + "text_projection", (tw, a.embed_dim), _normal(tw ** -0.5), F),
        ParamSpec(prefix + "logit_scale", (), _const(math.log(1 / 0.07)), F),
        ParamSpec(prefix + "visual.class_embedding", (vw,), _normal(s), F),
        ParamSpec(prefix + "visual.positional_embedding", (ntok, vw), _normal(s), F),
        ParamSpec(prefix + "visual.proj", (vw, a.embed_dim), _normal(s), F),
        ParamSpec(prefix + "visual.conv1.weight", (vw, 3, a.vision_patch_size, a.vision_patch_size),
                  _linear_weight(3 * a.vision_patch_size ** 2), F),
    ]
    out += _ln(prefix + "visual.ln_pre.", vw, F)
    for i in range(a.vision_layers):
        p = f"{prefix}visual.transformer.resblocks.{i}."
        out += [ParamSpec(p + "attn.in_proj_weight", (3 * vw, vw), _xavier(vw, 3 * vw), F),
                ParamSpec(p + "attn.in_proj_bias", (3 * vw,), _zeros, F),
                ParamSpec(p + "attn.out_proj.weight", (vw, vw), _linear_weight(vw), F),
                ParamSpec(p + "attn.out_proj.bias", (vw,), _zeros, F)]
        out += _ln(p + "ln_1.", vw, F)
        out += _linear(p + "mlp.c_fc.", 4 * vw, vw, F)
        out += _linear(p + "mlp.c_proj.", vw, 4 * vw, F)
        out += _ln(p + "ln_2.", vw, F)
    out += _ln(prefix + "visual.ln_post.", vw, F)
    proj_std = (tw ** -0.5) * ((2 * a.transformer_layers) ** -0.5)
    for i in range(a.transformer_layers):
        p = f"{prefix}transformer.resblocks.{i}."
        out += [ParamSpec(p + "attn.in_proj_weight", (3 * tw, tw), _normal(tw ** -0.5), F),
                ParamSpec(p + "attn.in_proj_bias", (3 * tw,), _zeros, F),
                ParamSpec(p + "attn.out_proj.weight", (tw, tw), _normal(proj_std), F),
                ParamSpec(p + "attn.out_proj.bias", (tw,), _zeros, F)]
        out += _ln(p + "ln_1.", tw, F)
        out += [ParamSpec(p + "mlp.c_fc.weight", (4 * tw, tw), _normal((2 * tw) ** -0.5), F),
                ParamSpec(p + "mlp.c_fc.bias", (4 * tw,), _linear_weight(tw), F),
                ParamSpec(p + "mlp.c_proj.weight", (tw, 4 * tw), _normal(proj_std), F),
                ParamSpec(p + "mlp.c_proj.bias", (tw,), _linear_weight(4 * tw), F)]
        out += _ln(p + "ln_2.", tw, F)
    out += [ParamSpec(prefix + "token_embedding.weight", (a.vocab_size, tw), _normal(0.02), F)]
    out += _ln(prefix + "ln_final.", tw, F)
    return out


def pooling_specs(prefix: str, input_dim: int, output_dim: int, layers: int, readouts: int) -> List[ParamSpec]:
    """AttentionPooling tensors (``otter/policy/transformer.py:162-216``)."""
    d = output_dim // readouts
    assert d * readouts == output_dim, "Output dim must be divisible by num readouts"
    out = [ParamSpec(prefix + "query", (1, readouts, d), _normal(1.0))]
    out += _ln(prefix + "layer_norm.", d)
    for i in range(layers):
        p = f"{prefix}blocks.{i}."
        out += [ParamSpec(p + "attention.q_proj_weight", (d, d), _xavier(d, d)),
                ParamSpec(p + "attention.k_proj_weight", (d, input_dim), _xavier(input_dim, d)),
                ParamSpec(p + "attention.v_proj_weight", (d, input_dim), _xavier(input_dim, d)),
                ParamSpec(p + "attention.in_proj_bias", (3 * d,), _zeros),
                ParamSpec(p + "attention.out_proj.weight", (d, d), _linear_weight(d)),
                ParamSpec(p + "attention.out_proj.bias", (d,), _zeros)]
        out += _linear(p + "mlp.fc1.", output_dim, d)
        out += _linear(p + "mlp.fc2.", d, output_dim)
        out += _ln(p + "q_layer_norm.", d)
        out += _ln(p + "kv_layer_norm.", input_dim)
        out += _ln(p + "layer_norm.", d)
    return out


def proprio_specs(prefix: str, inp: int, hidden: int, out_dim: int) -> List[ParamSpec]:
    """ProprioceptionEncoder tensors (``otter/policy/models.py:38-48``)."""
    return (_linear(prefix + "mlp.0.", hidden, inp) + _linear(prefix + "mlp.2.", hidden, hidden)
            + _linear(prefix + "mlp.4.", out_dim, hidden))


def action_head_specs(prefix: str, dim: int, action_dim: int, horizon: int) -> List[ParamSpec]:
    """ActionHead tensors (``otter/policy/models.py:53-62``)."""
    return _linear(prefix + "mlp.0.", 2 * dim, dim) + _linear(prefix + "mlp.2.", action_dim * horizon, 2 * dim)


def policy_specs(prefix: str, dim: int, heads: int, layers: int, expansion: int) -> List[ParamSpec]:
    """CausalTransformer tensors (``otter/policy/transformer.py:53-151``)."""
    if dim % heads != 0:
        raise ValueError(f"Hidden size {dim} is not divisible by number of attention heads {heads}")
    inner = dim * expansion
    out: List[ParamSpec] = []
    for i in range(layers):
        p = f"{prefix}layers.{i}."
        for name in ("query", "key", "value", "out"):
            out += _linear(p + f"attention.{name}.", dim, dim)
        out += _ln(p + "pre_ln.", dim)
        out += _ln(p + "post_ln.", dim)
        out += _linear(p + "mlp.0.", inner, dim)
        out += _linear(p + "mlp.2.", dim, inner)
    return out


def otter_specs(mc: ModelConfig, sc: SharedConfig, arch: Optional[ClipArch] = None) -> List[ParamSpec]:
    """All tensors of an OTTER checkpoint, in ``state_dict`` order of the reference's constructor."""
    a = arch or clip_arch(mc.clip_model)
    cams = len(sc.camera_keys)
    assert mc.vision_pooling_output_dim % cams == 0, \
        "Vision pooling output dim must be divisible by number of cameras"
    n_patches = (sc.image_size // a.vision_patch_size) ** 2
    dim = mc.transformer_dim
    out = clip_specs(a)
    out += pooling_specs("text_pooling.", a.embed_dim, mc.text_pooling_output_dim, mc.pooling_layers, mc.num_readouts)
    for c in range(cams):
        out += [ParamSpec(f"visual_extraction.{c}.temperature", (), _const(0.07)),
                ParamSpec(f"visual_extraction.{c}.pos_emb", (n_patches, a.embed_dim),
                          _fixed(sincos_position_embedding(n_patches, a.embed_dim)), "buffer")]
    for c in range(cams):
        out += pooling_specs(f"vision_poolings.{c}.", a.embed_dim, mc.vision_pooling_output_dim // cams,
                             mc.pooling_layers, mc.num_readouts)
    out += proprio_specs("proprio_encoder.", mc.proprio_input_dim, mc.proprio_hidden_dim, mc.proprio_output_dim)
    f_t = mc.text_pooling_output_dim + mc.vision_pooling_output_dim + mc.proprio_output_dim
    out += _linear("input_projection.", dim, f_t)
    out += [ParamSpec("pe.pe", (1, sc.seq_length, dim), _fixed(sinusoidal_table(sc.seq_length, dim)), "buffer")]
    out += policy_specs("policy.", dim, mc.transformer_heads, mc.transformer_layers, mc.transformer_expansion_factor)
    out += action_head_specs("action_head.", dim, mc.action_dim, sc.action_horizon)
    return out


# ---------------------------------------------------------------------------- module tree

class Node(nn.Module):
    """A name in the checkpoint tree.  Holds parameters / buffers / child nodes, computes nothing."""

    def __getitem__(self, idx) -> "Node":
        return self._modules[str(idx)]

    def __len__(self) -> int:
        return len(self._modules)

    def __iter__(self) -> Iterator["Node"]:
        return iter(self._modules.values())


def attach(root: nn.Module, spec: ParamSpec, gen: torch.Generator) -> None:
    parts = spec.key.split(".")
    node = root
    for name in parts[:-1]:
        child = node._modules.get(name)
        if child is None:
            child = Node()
            node.add_module(name, child)
        node = child
    value = spec.init(tuple(spec.shape), gen).to(torch.float32)
    assert tuple(value.shape) == tuple(spec.shape), (spec.key, value.shape, spec.shape)
    if spec.kind == "buffer":
        node.register_buffer(parts[-1], value)
    else:
        node.register_parameter(parts[-1], nn.Parameter(value, requires_grad=(spec.kind == "param")))


def random_state_dict(specs: List[ParamSpec], seed: int = 0, outlier_channels: int = 0, outlier_scale: float = 60.0) -> Dict[str, Tensor]:
    """A random-init checkpoint with the reference's initial distributions (not its RNG stream).

    ``outlier_channels`` > 0 mimics what trained CLIP towers look like and N(0, sigma) weights do not: that many channels
    of each tower's residual stream carry values ``outlier_scale`` times larger than the rest (the rows of every
    ``out_proj`` / ``c_proj`` that write them are scaled up, bias included) while the LayerNorm gains that read them are
    scaled down by the same factor.  Exercises range and rounding of a 16-bit residual stream and of LayerNorm folded into
    16-bit weights."""
    g = torch.Generator().manual_seed(seed)
    sd = {s.key: s.init(tuple(s.shape), g).to(torch.float32) for s in specs}
    if outlier_channels > 0:
        for tower in ("clip_model.visual.transformer.", "clip_model.transformer."):
            keys = [k for k in sd if k.startswith(tower)]
            if not keys:
                continue
            width = sd[tower + "resblocks.0.ln_1.weight"].numel()
            ch = torch.randperm(width, generator=g)[:outlier_channels]
            for k in keys:
                if k.endswith(("attn.out_proj.weight", "mlp.c_proj.weight", "attn.out_proj.bias", "mlp.c_proj.bias")):
                    sd[k][ch] *= outlier_scale
                    if k.endswith(".bias"):
                        sd[k][ch] += 0.5 * outlier_scale * torch.sign(torch.randn(len(ch), generator=g))   # a steady drift, as in trained towers
                elif k.endswith(("ln_1.weight", "ln_2.weight")):
                    sd[k][ch] /= outlier_scale
            for k in ("clip_model.visual.ln_post.weight", "clip_model.ln_final.weight"):
                if k in sd and sd[k].numel() == width:
                    sd[k][ch] /= outlier_scale
    return sd
