"""Loading published CLIP weights into an :class:`otter_amd.otter.OTTER` (``clip.load(name)`` at ``otter/policy/otter.py:62``).

The reference downloads the OpenAI archive by name; here the caller supplies a file (there is no network in the
deployment this package targets).  Accepted: a ``state_dict`` file of TENSORS ONLY in the OpenAI key layout
(``visual.conv1.weight``, ``transformer.resblocks.0.attn.in_proj_weight``, ``token_embedding.weight`` ...), fp16 or fp32,
optionally wrapped as ``{"state_dict": ...}``, or such a dict in memory.  The file is read with
``torch.load(weights_only=True)``: nothing in it is executed.  The OpenAI download itself is a TorchScript archive,
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
    sd = torch.load(path, map_location="cpu", weights_only=True)       # raises on anything but plain tensors / containers
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}


def load_clip_weights(model, source: Union[str, Dict[str, torch.Tensor]]) -> List[str]:
    """Copy CLIP tensors into ``model.clip_model.*`` (converted to the parameter's dtype).  Every CLIP tensor of the model
    must be present with the right shape; returns the keys of ``source`` that were not used."""
    sd = clip_state_dict_from_file(source) if isinstance(source, str) else dict(source)
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
