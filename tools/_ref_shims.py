"""Import shims that let the reference's hot-path modules be imported in the build container.

Only ``tools/gen_golden.py`` uses this (it never runs on the GPU box and never ships
reference source: it emits data fixtures).  The reference's modules import packages that
are not installed here (``clip``, ``timm``, ``torchvision``, ``tyro``, the TensorFlow data
layer); each shim below supplies just the names the hot-path files touch:

* ``otter.util.args``        -> ``ModelConfig`` / ``SharedConfig`` (field names and defaults of
                                 ``otter/util/args.py:121-178,219-264``), avoiding ``args.py:5``'s
                                 import of the TF/dlimp dataset code.
* ``timm.layers.mlp.Mlp``    -> fc1 -> GELU -> fc2 (the only timm class ``transformer.py:6`` uses).
* ``timm.data.constants``    -> the four mean/std tuples ``vision_tf.py:4`` reads.
* ``timm.data.transforms_factory`` / ``torchvision`` -> empty placeholders (imported, unused on the path).
* ``clip``                   -> an ``nn.Module`` with the OpenAI CLIP module tree (so the reference's
                                 forward hooks attach where ``otter.py:141-146`` expects) built at
                                 random init from a named architecture; no download.
"""
from __future__ import annotations

import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, "/root/repo")
from otter_amd.config import ModelConfig, SharedConfig, clip_arch  # noqa: E402


def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ------------------------------------------------------------------ timm / torchvision / args

class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


# ------------------------------------------------------------------ clip (OpenAI module tree)

class _LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


class _QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class _Block(nn.Module):
    def __init__(self, d, heads, mask=None, act=_QuickGELU):
        super().__init__()
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_1 = _LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d, 4 * d)), ("gelu", act()),
                                              ("c_proj", nn.Linear(4 * d, d))]))
        self.ln_2 = _LayerNorm(d)
        self.attn_mask = mask

    def forward(self, x):
        m = self.attn_mask.to(dtype=x.dtype, device=x.device) if self.attn_mask is not None else None
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=m)[0]
        return x + self.mlp(self.ln_2(x))


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads, mask=None, act=_QuickGELU):
        super().__init__()
        self.resblocks = nn.Sequential(*[_Block(width, heads, mask, act) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class _Vision(nn.Module):
    def __init__(self, res, patch, width, layers, heads, out_dim, act):
        super().__init__()
        self.output_dim = out_dim
        self.conv1 = nn.Conv2d(3, width, patch, patch, bias=False)
        s = width ** -0.5
        self.class_embedding = nn.Parameter(s * torch.randn(width))
        self.positional_embedding = nn.Parameter(s * torch.randn((res // patch) ** 2 + 1, width))
        self.ln_pre = _LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads, act=act)
        self.ln_post = _LayerNorm(width)
        self.proj = nn.Parameter(s * torch.randn(width, out_dim))

    def forward(self, x):
        x = self.conv1(x)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        cls = self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(x).permute(1, 0, 2)
        x = self.transformer(x).permute(1, 0, 2)
        x = self.ln_post(x[:, 0, :])
        return x @ self.proj


class _CLIP(nn.Module):
    def __init__(self, a):
        super().__init__()
        act = _QuickGELU if a.activation == "quick_gelu" else nn.GELU
        self.context_length = a.context_length
        self.visual = _Vision(a.image_resolution, a.vision_patch_size, a.vision_width, a.vision_layers,
                              a.vision_heads, a.embed_dim, act)
        mask = torch.full((a.context_length, a.context_length), float("-inf")).triu_(1)
        self.transformer = _Transformer(a.transformer_width, a.transformer_layers, a.transformer_heads, mask, act)
        self.vocab_size = a.vocab_size
        self.token_embedding = nn.Embedding(a.vocab_size, a.transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(a.context_length, a.transformer_width).normal_(std=0.01))
        self.ln_final = _LayerNorm(a.transformer_width)
        self.text_projection = nn.Parameter(torch.empty(a.transformer_width, a.embed_dim).normal_(std=a.transformer_width ** -0.5))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        return self.visual(image.type(self.dtype))

    def encode_text(self, text):
        x = self.token_embedding(text.long()).type(self.dtype) + self.positional_embedding.type(self.dtype)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = self.ln_final(x).type(self.dtype)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection


class _SimpleTokenizer:
    def __init__(self):
        self.encoder = {"<|startoftext|>": 49406, "<|endoftext|>": 49407}


def _clip_load(name, *a, **k):
    return _CLIP(clip_arch(name)), None


def install():
    import PIL.Image  # noqa: F401  (vision_tf.py:60 annotates with PIL.Image without importing the submodule)
    _module("otter.util.args", ModelConfig=ModelConfig, SharedConfig=SharedConfig)
    _module("timm")
    _module("timm.layers")
    _module("timm.layers.mlp", Mlp=Mlp)
    _module("timm.data")
    _module("timm.data.constants",
            IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225),
            OPENAI_CLIP_MEAN=(0.48145466, 0.4578275, 0.40821073), OPENAI_CLIP_STD=(0.26862954, 0.26130258, 0.27577711))
    _module("timm.data.transforms_factory", transforms_noaug_train=None)
    _module("torchvision")
    clip = _module("clip", load=_clip_load)
    st = _module("clip.simple_tokenizer", SimpleTokenizer=_SimpleTokenizer)
    clip.simple_tokenizer = st
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
