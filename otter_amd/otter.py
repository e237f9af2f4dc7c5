"""``OTTER``: the reference's model object (``otter/policy/otter.py:40-383``) with the compute on the MI355X.

Same constructor arguments, method names, attributes and ``state_dict`` layout as the reference class, so a
checkpoint written by the reference's trainer loads with ``load_state_dict(strict=True)`` and callers such as
``OtterInference`` / ``script/droid_inference.py`` find the attributes they read.  The module tree only *holds*
the tensors; the arithmetic runs in hand-written HIP kernels (:mod:`otter_amd.device_model`).  There is no CPU
or PyTorch implementation behind these methods: calling them without the HIP library and a GPU raises.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops
from .config import ModelConfig, SharedConfig, clip_arch
from .device_model import ClipDevice, OtterDevice
from .engine import BatchedRollout
from .params import Node, attach, otter_specs
from .tokenizer import EOT_TOKEN, SOT_TOKEN, tokenize

Tensor = torch.Tensor


def create_text_mask(text: Tensor, sot_token: int, eot_token: int, first_k_tokens: int) -> Tensor:
    """True at start-of-text, end-of-text and padding (id 0) positions, cut to the first k tokens
    (``otter/policy/otter.py:13-38``).  Host-side integer logic on a (77,) or (B, 77) id tensor."""
    assert text.dim() in (1, 2), "text must be either 1D or 2D"
    mask = (text == sot_token) | (text == eot_token) | (text == 0)
    return mask[..., :first_k_tokens]


class OTTER(nn.Module):
    """Full OTTER model with temporal sequence handling (device execution)."""

    def __init__(self, model_config: ModelConfig, shared_config: SharedConfig):
        super().__init__()
        mc, sc = model_config, shared_config
        self.clip_model_name = mc.clip_model
        self.clip_arch = clip_arch(mc.clip_model)
        self.proprio_input_dim = mc.proprio_input_dim
        self.action_dim = mc.action_dim
        self.first_k_tokens = mc.first_k_tokens
        self.num_readouts = mc.num_readouts
        self.pool_true_text = mc.pool_true_text
        self.pooling_heads = mc.pooling_heads
        self.transformer_heads = mc.transformer_heads
        self.action_horizon = sc.action_horizon
        self.use_mse = sc.mse_loss
        self.camera_keys = sorted(sc.camera_keys)
        self.seq_length = sc.seq_length
        self.image_size = sc.image_size
        self.sot_token: int = SOT_TOKEN
        self.eot_token: int = EOT_TOKEN
        self.visual_patch_size = self.clip_arch.vision_patch_size
        self.num_img_patches = (self.image_size // self.visual_patch_size) ** 2
        self.proprio_output_dim = mc.proprio_output_dim
        self.f_t_dim = mc.text_pooling_output_dim + mc.vision_pooling_output_dim + mc.proprio_output_dim

        for spec in otter_specs(mc, sc, self.clip_arch):
            attach(self, spec, None)

        # 16-bit operand type of the CLIP vision tower's MFMAs.  fp16 (default) has 3 more mantissa bits than bf16 at the
        # same MFMA rate and represents the published CLIP weights exactly; bf16 trades that for fp32's range.
        self.clip_compute_dtype = torch.float16
        # Residual stream of the towers: None = fp16 with fp16 operands (the reference's GPU path carries fp16 too), fp32 with
        # bf16 operands; torch.float32 forces the fp32 stream (+2.4 % step time, 4x smaller rounding error in the features).
        self.clip_stream_dtype = None
        # Operand type of the text tower: None = fp16 whatever the vision tower uses (bf16 rounding of the text tower alone
        # costs more action error than the tolerance allows; the tower runs once per prompt and is 0.8 % of the work).
        self.clip_text_dtype = None
        self._dev: Optional[OtterDevice] = None
        self._dev_key = None
        self._clip_dev: Optional[ClipDevice] = None
        self._clip_key = None
        self._tail = None
        self._rollout: Optional[BatchedRollout] = None
        self._param_groups = None
        self._carry_window = None
        self.check_weights_by_value = False       # opt-in: a value checksum of the weights in the staleness key (one device sync per forward)
        self.cache_size = 0

    # ------------------------------------------------------------------ device state
    def _apply(self, fn, *a, **k):
        self._param_groups = None
        self._dev = None
        self._clip_dev = None
        self._tail = None
        self._rollout = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._param_groups = None
        self._dev = None
        self._clip_dev = None
        self._rollout = None
        return super().load_state_dict(*a, **k)

    def _weights_key(self, trainable: bool):
        """Changes whenever a tensor of the group is written in place through the tensor itself (optimizer step, ``p.copy_``,
        ``p.mul_``: the version counter) or re-pointed (``p.data = other``: the storage address); moves and
        ``load_state_dict`` drop the packed copies directly.  Writes THROUGH ``p.data`` / ``p.detach()`` (``p.data.copy_()``,
        an EMA ``p.data.mul_()``) or through raw pointers change neither: call :meth:`invalidate_device_weights` after them,
        or set ``check_weights_by_value`` to have every forward compare a fingerprint of the values (per-tensor sum and sum
        of squares: a heuristic that catches optimizer-style writes, not a proof of equality).  Buffers that the device
        model packs (``pe.pe``, ``visual_extraction.*.pos_emb``) are part of the trainable group's key.  Called on every
        forward: the tensor lists are cached (the module tree is fixed after construction) so that the check costs two
        attribute reads per tensor."""
        if self._param_groups is None:
            named = list(self.named_parameters())
            self._param_groups = ([p for n, p in named if n.startswith("clip_model.")],
                                  [p for n, p in named if not n.startswith("clip_model.")] + [b for _, b in self.named_buffers()])
        group = self._param_groups[1 if trainable else 0]
        key = tuple(p._version for p in group) + tuple(p.data_ptr() for p in group)
        if self.check_weights_by_value:
            # per tensor: sum and sum of squares in fp64, weighted by the tensor's index -- a heuristic fingerprint, not a
            # comparison of every value: it sees scaling, sign flips, zero-mean perturbations and writes to one tensor that
            # a plain sum over everything would miss, but not, e.g., two rows of one tensor swapping places
            vals = [torch.stack([p.detach().double().sum(), p.detach().double().square().sum()]) * (i + 1) for i, p in enumerate(group)]
            key += tuple(torch.stack(vals).sum(0).tolist())
        return key

    def invalidate_device_weights(self, towers: bool = True, keep_history: bool = True) -> None:
        """Drop the packed device copies of the weights (and the captured graphs built on them): the next forward repacks
        from the module's tensors.  For writers the automatic check cannot see: in-place writes through ``p.data`` /
        ``p.detach()``, raw-pointer kernels (the device AdamW of :class:`otter_amd.trainer.Trainer`).  ``towers=False``
        keeps the packed CLIP towers (0.86 GB; only the trainable modules changed).  ``keep_history`` carries the running
        episode's window of policy inputs over to the new weights (also across several invalidations with no inference
        call between them: the window taken at the first one is kept)."""
        if not keep_history:
            self._carry_window = None
        elif self._rollout is not None:
            self._carry_window = self._window_snapshot()
        # else: no live rollout -- whatever an earlier invalidation carried stays
        self._dev = None
        if towers:
            self._clip_dev = None
        if self._tail is not None:
            self._tail._dims_key = None                # packed buffer tables of the training tail
        self._rollout = None

    def _device(self) -> torch.device:
        device = self.input_projection.weight.device
        if device.type != "cuda":
            raise ops.OtterHipError(
                f"OTTER is on {device}; move it to the GPU first (model.to('cuda')). The hot path has no CPU implementation.")
        return device

    def clip_device(self) -> ClipDevice:
        """Packed weights of the frozen CLIP towers (rebuilt only when a CLIP tensor or the operand type changes)."""
        device = self._device()
        key = (self._weights_key(False), self.clip_compute_dtype, self.clip_stream_dtype, self.clip_text_dtype, device)
        if self._clip_dev is None or self._clip_key != key:
            sd = {k: v for k, v in self.state_dict().items() if k.startswith("clip_model.")}
            self._clip_dev = ClipDevice(sd, self.clip_arch, self.first_k_tokens, device, self.clip_compute_dtype,
                                        stream_dtype=self.clip_stream_dtype, text_dtype=self.clip_text_dtype)
            self._clip_key = key
            self._dev = None
        return self._clip_dev

    def device_model(self) -> OtterDevice:
        """Packed device weights, rebuilt when the module moved or a trainable parameter changed."""
        device = self._device()
        clip = self.clip_device()
        key = self._weights_key(True)
        if self._dev is None or self._dev_key != key:
            if self._rollout is not None and self._carry_window is None:
                self._carry_window = self._window_snapshot()      # a weight changed mid-episode: the history survives the repack
            self._dev = OtterDevice(
                dict(self.state_dict()), self.clip_arch, first_k=self.first_k_tokens, pooling_heads=self.pooling_heads,
                transformer_heads=self.transformer_heads, num_cameras=len(self.camera_keys), seq_length=self.seq_length,
                action_horizon=self.action_horizon, action_dim=self.action_dim, pool_true_text=self.pool_true_text, device=device,
                clip_dtype=self.clip_compute_dtype, clip=clip)
            self._dev_key = key
            self._rollout = None
            if self._tail is not None:
                self._tail._dims_key = None               # the training tail packs the same buffers (pe, pos_emb): repack them too
        return self._dev

    # ------------------------------------------------------------------ rollout cache (otter.py:153-175)
    def _window_snapshot(self) -> Optional[Tensor]:
        if self._rollout is None or self.cache_size == 0:
            return None
        return self._rollout.window[:, : self.cache_size].clone()

    def _get_rollout(self) -> BatchedRollout:
        dev = self.device_model()
        if self._rollout is None:
            self._rollout = BatchedRollout(dev, 1)
            self.cache_size = 0
            carried, self._carry_window = self._carry_window, None
            if carried is not None:
                # The policy inputs of the running episode are kept; they were produced by the OLD encoder weights, as
                # they would be in the reference (its cache holds inputs, otter.py:158-175).  K/V rows are recomputed.
                self.update_cache(carried)
        return self._rollout

    def reset_cache(self):
        """Reset the transformer input cache."""
        self._carry_window = None
        if self._rollout is not None:
            self._rollout.reset()
        self.cache_size = 0

    @property
    def cached_transformer_input(self) -> Optional[Tensor]:
        """(1, cache_size, transformer_dim) policy inputs of the running episode, or None (``otter.py:149-151``)."""
        if self._rollout is None or self.cache_size == 0:
            return self._carry_window
        return self._rollout.window[:, : self.cache_size]

    @cached_transformer_input.setter
    def cached_transformer_input(self, value: Optional[Tensor]) -> None:
        # the reference exposes a plain attribute: ``= None`` clears the cache, a tensor replaces its contents
        self.reset_cache()
        if value is not None:
            if value.dim() != 3 or value.shape[0] != 1:
                raise ValueError(f"cached_transformer_input must be (1, T, dim), got {tuple(value.shape)}")
            self.update_cache(value[:, -self.seq_length:])

    # ------------------------------------------------------------------ helpers
    def _images_nchw(self, images: Tensor) -> Tuple[Tensor, int, int]:
        """(B, T, cams, 3, H, W) uint8 / float -> ((B*T*cams, 3, H, W) uint8 or fp32 normalised, B, T)."""
        b, t, cams = images.shape[:3]
        if cams != len(self.camera_keys):
            raise ValueError(f"expected {len(self.camera_keys)} cameras, got {cams}")
        flat = images.reshape(b * t * cams, *images.shape[3:])
        if flat.dtype != torch.uint8:
            flat = flat.float()
        return flat.contiguous(), b, t

    def _text_feats(self, text: Tensor, dev: Optional[OtterDevice] = None) -> Tensor:
        dev = dev if dev is not None else self.device_model()
        ids = text.detach().to("cpu", torch.int32)
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.clip_arch.vocab_size):
            raise ValueError("token id outside the CLIP vocabulary")
        return dev.clip.text_tokens_cached(ids.view(-1, ids.shape[-1]))

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def extract_clip_features(self, images: Tensor, text: Tensor) -> Tuple[List[Tensor], Tensor]:
        """``otter.py:177-222``: L2-normalised patch tokens per camera and text tokens, fp32."""
        dev = self.device_model()
        flat, b, t = self._images_nchw(images)
        cams = len(self.camera_keys)
        tokens, p, d = dev.clip.vision_tokens(flat)
        pf = ops.l2norm_rows(tokens.view(-1, d)).view(b * t, cams, p + 1, d)[:, :, 1:]
        tf_raw = self._text_feats(text)
        tf = ops.l2norm_rows(tf_raw.view(-1, d)).view(b, self.first_k_tokens, d)
        tf = tf.repeat_interleave(t, dim=0)
        return [pf[:, i] for i in range(cams)], tf

    @torch.no_grad()
    def forward_encoder(self, images: Tensor, text: Tensor, proprio: Tensor, text_mask: Optional[Tensor] = None) -> Tensor:
        """``otter.py:224-267``: (B, T, cams, C, H, W), (B, 77), (B, T, P_in) -> (B, T, transformer_dim)."""
        dev = self.device_model()
        flat, b, t = self._images_nchw(images)
        tf_raw = self._text_feats(text)
        prop = proprio.reshape(b * t, -1).to(dev.device, torch.float32).contiguous()
        x = dev.encode(flat.to(dev.device), tf_raw, prop, t, text_mask if self.pool_true_text else None)
        return x.view(b, t, -1)

    @torch.no_grad()
    def forward_actions(self, images: Tensor, text: Tensor, proprio: Tensor, text_mask: Optional[Tensor] = None) -> Tensor:
        """``otter.py:269-291`` -> (B, T, action_horizon, action_dim)."""
        dev = self.device_model()
        flat, b, t = self._images_nchw(images)
        tf_raw = self._text_feats(text)
        prop = proprio.reshape(b * t, -1).to(dev.device, torch.float32).contiguous()
        return dev.actions(flat.to(dev.device), tf_raw, prop, b, t, text_mask if self.pool_true_text else None)

    def update_cache(self, transformer_input: Tensor):
        """Append (1, T, transformer_dim) policy inputs to the rollout window (``otter.py:158-175``)."""
        ro = self._get_rollout()
        x = transformer_input.to(ro.dev.device, torch.float32)
        for i in range(x.shape[1]):
            nxt = ro._cur ^ 1
            ops.window_append(ro._win[ro._cur], x[:, i].contiguous(), ro._win[nxt], ro._lens[ro._cur], ro._lens[nxt])
            ro._cur = nxt
            ro.lens_host[:] = min(int(ro.lens_host[0]) + 1, ro.S)
        ro.invalidate_kv()            # K/V history was not produced for these rows: recompute the window
        self.cache_size = int(ro.lens_host[0])

    @torch.no_grad()
    def forward_inference(self, images: Tensor, text: Tensor, proprio: Tensor, text_mask: Optional[Tensor] = None) -> Tensor:
        """One rollout step of a single episode (``otter.py:293-337``).

        images (cams, C, H, W) uint8 or normalised float, text (77,) ids, proprio (P_in,) -> (horizon, action_dim).
        """
        ro = self._get_rollout()
        dev = ro.dev
        imgs = images if images.dtype == torch.uint8 else images.float()
        tf_raw = self._text_feats(text.view(1, -1), dev)
        mask = text_mask.view(1, -1) if (self.pool_true_text and text_mask is not None) else None
        out = ro.step(imgs.to(dev.device).contiguous(), tf_raw, proprio.view(1, -1).to(dev.device, torch.float32).contiguous(), mask)
        self.cache_size = int(ro.lens_host[0])
        return out.view(self.action_horizon, self.action_dim)

    def forward(self, images: Union[Tensor, dict], text: Union[Tensor, List[str]], proprio: Tensor,
                gt_actions: Optional[Tensor] = None) -> Tensor:
        """``otter.py:339-383``: actions (B, T, H, A), or the L1 / MSE loss when ``gt_actions`` is given."""
        if isinstance(images, dict):
            images = torch.stack([images[key] for key in self.camera_keys], dim=2)
            if images.shape[-1] == 3:
                images = images.permute(0, 1, 2, 5, 3, 4)
        if isinstance(text, list):
            text = tokenize(text)
        text_mask = create_text_mask(text, self.sot_token, self.eot_token, self.first_k_tokens) if self.pool_true_text else None
        # Every non-float dtype is normalised (otter.py:367-368).  uint8 images are normalised on the device inside the
        # patch-extraction kernel (vision_tf.py:35-45); other integer types hold pixel values too and take the same
        # route when they fit a byte, else the elementwise transform itself (/255, mean, std) on their device.
        if not torch.is_floating_point(images) and images.dtype != torch.uint8:
            if images.numel() == 0 or (int(images.min()) >= 0 and int(images.max()) <= 255):
                images = images.to(torch.uint8)
            else:
                from .vision_tf import clip_transform_host
                images = clip_transform_host(images)
        if gt_actions is None:
            return self.forward_actions(images, text, proprio, text_mask)
        return self.loss(images, text, proprio, gt_actions, text_mask)

    def loss(self, images: Tensor, text: Tensor, proprio: Tensor, gt_actions: Tensor, text_mask: Optional[Tensor] = None) -> Tensor:
        """L1 / MSE loss of the predicted chunks (``otter.py:378-382``) as a scalar tensor.  With autograd enabled it is a
        differentiable function of the trainable parameters (hand-written backward, :mod:`otter_amd.train`)."""
        from .train import TailLoss, TrainableTail
        if self._tail is None:
            self._tail = TrainableTail(self)
        tail = self._tail
        device = self._device()
        ids = text.detach().to("cpu", torch.int32).view(-1, text.shape[-1])
        imgs = images.to(device)
        names = tail.trainable_names()                       # frozen parameters take part in the forward pass only
        if torch.is_grad_enabled() and names:
            return TailLoss.apply(tail, imgs, ids, proprio, gt_actions, text_mask, names, *[tail.params[n] for n in names])
        with torch.no_grad():
            return tail.forward_backward(imgs, ids, proprio, gt_actions, text_mask, None).view(())

    def __repr__(self):
        n = lambda it: sum(p.numel() for p in it)          # noqa: E731
        return (f"OTTER(clip={self.clip_model_name}, cameras={self.camera_keys}, seq_length={self.seq_length}, "
                f"action_horizon={self.action_horizon}, action_dim={self.action_dim}, params={n(self.parameters()):,}, "
                f"trainable={n(p for p in self.parameters() if p.requires_grad):,})")
