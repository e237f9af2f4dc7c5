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


# ------------------------------------------------------------------------------------------- device-side crop + resize
_PRECISION_BITS = 32 - 8 - 2          # PIL's 8-bit resampler keeps 22 fractional bits


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size: int, out_size: int):
    """Source windows and fixed-point weights of one axis of PIL's bicubic resize from ``in_size`` to ``out_size``
    samples (the arithmetic of Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc``: window = centre +/- 2 source
    pixels, widened by the scale factor when shrinking (antialiasing); weights normalised, then rounded to 22
    fractional bits, half away from zero).  Returns (bounds int32 (out, 2), coef int32 (out, ksize), ksize)."""
    import math
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        total = sum(w)
        if total != 0.0:
            w = [v / total for v in w]
        for x, v in enumerate(w):
            coef[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coef, ksize


_COEFF_CACHE = {}


def crop_resize_device(frames: torch.Tensor, out_size: int) -> torch.Tensor:
    """uint8 camera frames (n, H, W, 3) on the GPU -> uint8 (n, 3, out_size, out_size): centre crop to a square and bicubic
    resize, bit-identical to ``crop2square(img).resize((S, S), Image.BICUBIC)`` on the host, in two small kernels."""
    from . import ops
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3 or not frames.is_contiguous():
        raise ValueError("crop_resize_device: expected contiguous uint8 frames of shape (n, H, W, 3)")
    if frames.device.type != "cuda":
        raise ops.OtterHipError("crop_resize_device: frames must be on the GPU (the host path is PIL)")
    n, h, w, _ = frames.shape
    side = min(h, w)
    y0, x0 = ((h - w) // 2, 0) if h > w else (0, (w - h) // 2)
    key = (side, out_size, frames.device)
    if key not in _COEFF_CACHE:
        b, c, k = pil_bicubic_coeffs(side, out_size)
        _COEFF_CACHE[key] = (torch.from_numpy(b).to(frames.device), torch.from_numpy(c).to(frames.device), k)
    b, c, k = _COEFF_CACHE[key]
    tmp = torch.empty(n, side, out_size, 3, dtype=torch.uint8, device=frames.device)
    out = torch.empty(n, 3, out_size, out_size, dtype=torch.uint8, device=frames.device)
    ops._lib.call("otter_resize_bicubic_u8", ops._ptr(frames), n, h, w, y0, x0, side, side, ops._ptr(b), ops._ptr(c), k, ops._ptr(b), ops._ptr(c), k,
                  ops._ptr(tmp), ops._ptr(out), out_size, ops._stream())
    return out


def crop_resize_reference_arithmetic(frame: np.ndarray, out_size: int) -> np.ndarray:
    """The same two passes in numpy (H, W, 3) uint8 -> (3, S, S) uint8: documents the arithmetic and lets the CPU tests hold
    the coefficient tables against PIL without a GPU."""
    h, w, _ = frame.shape
    side = min(h, w)
    y0, x0 = ((h - w) // 2, 0) if h > w else (0, (w - h) // 2)
    crop = frame[y0:y0 + side, x0:x0 + side].astype(np.int64)
    b, c, _ = pil_bicubic_coeffs(side, out_size)

    def one_pass(img, axis_len_out):
        out = np.empty((img.shape[0], axis_len_out, 3), dtype=np.int64)
        for xx in range(axis_len_out):
            lo, cnt = b[xx]
            acc = (1 << (_PRECISION_BITS - 1)) + np.tensordot(img[:, lo:lo + cnt, :], c[xx, :cnt].astype(np.int64), axes=([1], [0]))
            out[:, xx, :] = np.clip(acc >> _PRECISION_BITS, 0, 255)
        return out
    horizontal = one_pass(crop, out_size)                                   # (side, S, 3)
    vertical = one_pass(horizontal.transpose(1, 0, 2), out_size)            # (S(x), S(y), 3)
    return vertical.transpose(2, 1, 0).astype(np.uint8)                     # (3, y, x)
