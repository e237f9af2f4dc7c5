"""Host-side image helpers of the rollout interface (``otter/policy/vision_tf.py``)."""
from __future__ import annotations

import numpy as np
import torch

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)     # timm.data.constants values read at vision_tf.py:4
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def crop2square(pil_img):
    """Centre crop to a square, flooring the offset (``vision_tf.py:60-67``)."""
    w, h = pil_img.size
    if h == w:
        return pil_img
    if h > w:
        top = (h - w) // 2
        return pil_img.crop((0, top, w, w + top))
    left = (w - h) // 2
    return pil_img.crop((left, 0, h + left, h))


def pil_to_uint8_chw(pil_img) -> torch.Tensor:
    """RGB PIL image -> uint8 (3, H, W).  The /255 and the CLIP mean/std normalisation that the reference applies
    on the host (``to_tensor`` + ``clip_transform``, ``otter_interface.py:106-107``) happen on the device."""
    arr = np.asarray(pil_img.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous()


def clip_transform_host(obs: torch.Tensor) -> torch.Tensor:
    """Host version of ``clip_transform`` (``vision_tf.py:28-52``) for callers that want normalised floats."""
    if not torch.is_floating_point(obs) or obs.max() > 1.0:
        obs = obs.float() / 255.0
    shape = [1] * (obs.dim() - 3) + [3, 1, 1]
    mean = torch.tensor(OPENAI_CLIP_MEAN, dtype=torch.float32, device=obs.device).view(shape)
    std = torch.tensor(OPENAI_CLIP_STD, dtype=torch.float32, device=obs.device).view(shape)
    return (obs - mean) / std
