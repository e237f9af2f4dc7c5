"""Loading published CLIP weights into an :class:`otter_amd.otter.OTTER` (``clip.load(name)`` at ``otter/policy/otter.py:62``).

The reference downloads the OpenAI archive by name; here the caller supplies a file (there is no network in the
deployment this package targets).  Accepted: a ``state_dict`` of TENSORS ONLY, fp16 / bf16 / fp32, as a ``torch.save`` file
(optionally wrapped as ``{"state_dict": ...}``), as a ``.safetensors`` file, or as a dict in memory, in one of two key layouts:

* the OpenAI layout (``visual.conv1.weight``, ``transformer.resblocks.0.attn.in_proj_weight``, ``token_embedding.weight`` ...),
  which is also what OpenCLIP's ``ViT-L-14`` checkpoints use (``open_clip_pytorch_model.bin`` / ``open_clip_model.safetensors``
  of ``datacomp_xl_s13b_b90k``, ``script/open_clip_visualization.py:25-38``; their ``attn_mask`` buffer is reported unused);
* the Hugging Face ``transformers.CLIPModel`` layout (``vision_model.encoder.layers.0.self_attn.q_proj.weight`` ...:
  ``model.safetensors`` / ``pytorch_model.bin`` of ``openai/clip-vit-large-patch14``), converted on the fly: q / k / v
  projections concatenated into ``in_proj``, the two projection matrices transposed.

``torch.save`` files are read with ``torch.load(weights_only=True)`` and safetensors files hold raw tensor bytes: nothing in
either is executed.  The OpenAI download itself is a TorchScript archive,
which cannot be opened that way and is deliberately NOT opened here (loading it runs code from the file); whoever
trusts that file converts it once, elsewhere, with ``torch.save(torch.jit.load(f).state_dict(), out)``.
An OTTER checkpoint needs none of this: it already contains the CLIP tensors.
"""
from __future__ import annotations

from typing import Dict, List, Union

import torch

# entries of the OpenAI archive that are not tensors of the model
_NOT_PARAMETERS = ("input_resolution", "context_length", "vocab_size")


def clip_state_dict_from_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path, device="cpu")
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)   # raises on anything but plain tensors / containers
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}


def is_transformers_layout(sd: Dict[str, torch.Tensor]) -> bool:
    return any(k.startswith(("vision_model.", "text_model.")) for k in sd)


def from_transformers_layout(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``transformers.CLIPModel.state_dict()`` -> the OpenAI key layout.  Same tensors: the attention's three projections
    are concatenated in q, k, v order (``nn.MultiheadAttention.in_proj_*``), ``visual_projection`` / ``text_projection``
    are ``nn.Linear`` weights there and plain matrices multiplied from the right here (transposed); buffers such as
    ``position_ids`` have no counterpart and are dropped."""
    out: Dict[str, torch.Tensor] = {}
    fixed = {
        "vision_model.embeddings.class_embedding": "visual.class_embedding",
        "vision_model.embeddings.patch_embedding.weight": "visual.conv1.weight",
        "vision_model.embeddings.position_embedding.weight": "visual.positional_embedding",
        "vision_model.pre_layrnorm.weight": "visual.ln_pre.weight",              # (sic: the upstream attribute name)
        "vision_model.pre_layrnorm.bias": "visual.ln_pre.bias",
        "vision_model.post_layernorm.weight": "visual.ln_post.weight",
        "vision_model.post_layernorm.bias": "visual.ln_post.bias",
        "text_model.embeddings.token_embedding.weight": "token_embedding.weight",
        "text_model.embeddings.position_embedding.weight": "positional_embedding",
        "text_model.final_layer_norm.weight": "ln_final.weight",
        "text_model.final_layer_norm.bias": "ln_final.bias",
        "logit_scale": "logit_scale",
    }
    for src, dst in fixed.items():
        if src in sd:
            out[dst] = sd[src]
    for src, dst in (("visual_projection.weight", "visual.proj"), ("text_projection.weight", "text_projection")):
        if src in sd:
            out[dst] = sd[src].t().contiguous()
    per_block = (("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"), ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj"),
                 ("self_attn.out_proj", "attn.out_proj"))
    for tower, root in (("vision_model", "visual.transformer"), ("text_model", "transformer")):
        i = 0
        while f"{tower}.encoder.layers.{i}.self_attn.q_proj.weight" in sd:
            h, o = f"{tower}.encoder.layers.{i}.", f"{root}.resblocks.{i}."
            for kind in ("weight", "bias"):
                out[o + "attn.in_proj_" + kind] = torch.cat([sd[h + f"self_attn.{x}_proj.{kind}"] for x in "qkv"], 0)
                for a, b in per_block:
                    out[o + b + "." + kind] = sd[h + a + "." + kind]
            i += 1
    return out


def load_clip_weights(model, source: Union[str, Dict[str, torch.Tensor]]) -> List[str]:
    """Copy CLIP tensors into ``model.clip_model.*`` (converted to the parameter's dtype).  Every CLIP tensor of the model
    must be present with the right shape; returns the keys of ``source`` that were not used."""
    sd = clip_state_dict_from_file(source) if isinstance(source, str) else dict(source)
    if is_transformers_layout(sd):
        sd = from_transformers_layout(sd)
    own = model.state_dict()
    used, unused = set(), []
    with torch.no_grad():
        for k, v in sd.items():
            key = k if k.startswith("clip_model.") else "clip_model." + k
            if key not in own or k in _NOT_PARAMETERS:
                unused.append(k)
                continue
            if tuple(own[key].shape) != tuple(v.shape):
                raise ValueError(f"{k}: shape {tuple(v.shape)} in the file, {tuple(own[key].shape)} in {model.clip_model_name}")
            own[key].copy_(v.to(own[key].dtype))
            used.add(key)
    missing = [k for k in own if k.startswith("clip_model.") and k not in used]
    if missing:
        raise KeyError(f"{len(missing)} CLIP tensors missing from the source, e.g. {missing[:3]}")
    model._clip_dev = None                              # packed device copies are stale
    model._dev = None
    model._rollout = None
    return unused
