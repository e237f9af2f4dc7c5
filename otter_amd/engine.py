"""Batched rollout engine: many concurrent episodes, GPU-resident history.

The reference serves one episode per model object and re-encodes the whole history window on every call
(``otter/policy/otter.py:293-337``).  Its cache holds the policy *inputs* (pre positional-embedding), keeps the
last ``seq_length`` of them and indexes the sinusoidal table by position inside the window.  Consequences the
engine relies on:

* while an episode has fewer than ``seq_length`` steps, step t sits at window position t forever and the
  causal mask isolates it from later steps, so per-layer K/V of earlier steps can be kept on the GPU and each
  call only computes the new position ("cached" mode);
* once the window is full every call drops the oldest input and all positions shift by one, so K/V of every
  remaining step change: the window is recomputed ("full" mode; at most ``seq_length`` tokens per episode).

Episodes advance in lock-step; ``reset(episode_ids)`` of a subset switches the engine to full mode with
per-episode lengths (rows past an episode's length are zero and, by causality, never influence earlier rows).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import ops
from .device_model import OtterDevice

Tensor = torch.Tensor

# A step of a few episodes is launch-bound (about 330 small kernels for the encoder of one observation): up to this many
# images per step the encoder is captured once into a HIP graph and replayed.  OTTER_HIP_GRAPH=0 keeps it eager.
GRAPH_MAX_IMAGES = 16
_USE_GRAPH = os.environ.get("OTTER_HIP_GRAPH", "1") != "0"


class _EncodeGraph:
    """``OtterDevice.encode`` for one fixed set of input shapes, captured into a HIP graph: static input buffers, one
    replay per step.  The weights' addresses are baked in, so the object lives and dies with its ``OtterDevice``."""

    @staticmethod
    def key_of(images: Tensor, text_feats: Tensor, proprio: Tensor, text_mask: Optional[Tensor]):
        return (tuple(images.shape), images.dtype, tuple(text_feats.shape), tuple(proprio.shape),
                None if text_mask is None else tuple(text_mask.shape))

    def __init__(self, dev: OtterDevice, images: Tensor, text_feats: Tensor, proprio: Tensor, text_mask: Optional[Tensor]):
        self.key = self.key_of(images, text_feats, proprio, text_mask)
        d = dev.device
        self.images, self.text, self.proprio = images.to(d).clone(), text_feats.to(d).clone(), proprio.to(d).clone()
        self.mask = None if text_mask is None else text_mask.to(d).clone()
        # the graph's own split-K scratch: its address is baked into the captured launches, and another graph may be
        # replayed on another stream at the same time
        self.ws = ops.new_gemm_workspace(d)
        cur = torch.cuda.current_stream(d)
        side = torch.cuda.Stream(device=d)
        side.wait_stream(cur)
        with torch.cuda.stream(side), ops.gemm_workspace(self.ws):     # once eagerly: LDS opt-ins and allocator pools exist before capture
            dev.encode(self.images, self.text, self.proprio, 1, self.mask)
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), ops.gemm_workspace(self.ws):
            self.out = dev.encode(self.images, self.text, self.proprio, 1, self.mask)

    def __call__(self, images: Tensor, text_feats: Tensor, proprio: Tensor, text_mask: Optional[Tensor]) -> Tensor:
        self.images.copy_(images, non_blocking=True)
        self.text.copy_(text_feats, non_blocking=True)
        self.proprio.copy_(proprio, non_blocking=True)
        if self.mask is not None:
            self.mask.copy_(text_mask, non_blocking=True)
        self.graph.replay()
        return self.out                            # overwritten by the next replay: consume it on this stream before then


class BatchedRollout:
    def __init__(self, dev: OtterDevice, n_episodes: int):
        self.dev, self.n = dev, int(n_episodes)
        self.S = dev.seq_length
        d = dev.device
        self._win = [torch.zeros(self.n, self.S, dev.dim, dtype=torch.float32, device=d) for _ in range(2)]
        self._lens = [torch.zeros(self.n, dtype=torch.int32, device=d) for _ in range(2)]
        self._cur = 0
        self._cache = dev.policy.new_cache(self.n, self.S)
        self.lens_host = np.zeros(self.n, dtype=np.int64)
        self._lockstep = True
        self._graph: Optional[_EncodeGraph] = None

    # ------------------------------------------------------------------ state
    @property
    def window(self) -> Tensor:
        """(n, seq_length, dim) policy inputs; episode e's valid rows are [:lens_host[e]]."""
        return self._win[self._cur]

    def reset(self, episode_ids: Optional[Sequence[int]] = None) -> None:
        if episode_ids is None:
            for w in self._win:
                w.zero_()
            for l in self._lens:
                l.zero_()
            self.lens_host[:] = 0
            self._lockstep = True
            return
        ids = torch.as_tensor(list(episode_ids), dtype=torch.long, device=self.dev.device)
        if ids.numel() == 0:
            return
        self._win[self._cur].index_fill_(0, ids, 0.0)
        self._lens[self._cur].index_fill_(0, ids, 0)
        self.lens_host[list(episode_ids)] = 0
        self._lockstep = bool(np.all(self.lens_host == 0))

    # ------------------------------------------------------------------ one observation per episode
    def step(self, images: Tensor, text_feats: Tensor, proprio: Tensor, text_mask: Optional[Tensor] = None) -> Tensor:
        """images (n*cams, 3, R, R); text_feats (n, K, D) fp32 un-normalised; proprio (n, P_in) fp32.

        Returns the action chunks, fp32 (n, horizon, action_dim), still on the device.
        """
        dev = self.dev
        if _USE_GRAPH and images.shape[0] <= GRAPH_MAX_IMAGES and not torch.cuda.is_current_stream_capturing():
            if self._graph is None or self._graph.key != _EncodeGraph.key_of(images, text_feats, proprio, text_mask):
                self._graph = _EncodeGraph(dev, images, text_feats, proprio, text_mask)
            fresh = self._graph(images, text_feats, proprio, text_mask)          # (n, dim), the graph's output buffer
        else:
            fresh = dev.encode(images, text_feats, proprio, 1, text_mask)        # (n, dim)
        nxt = self._cur ^ 1
        ops.window_append(self._win[self._cur], fresh, self._win[nxt], self._lens[self._cur], self._lens[nxt])
        self._cur = nxt
        was_full = self.lens_host >= self.S
        self.lens_host = np.minimum(self.lens_host + 1, self.S)
        win = self._win[self._cur]
        if self._lockstep and not was_full[0]:
            pos = int(self.lens_host[0]) - 1
            y = dev.policy.run(win, self._cache, 1, pos)                         # (n, dim): only the new position
            out = dev.policy.head(y)
        else:
            y = dev.policy.run(win, self._cache, self.S, 0)                      # whole window, (n*S, dim)
            if self._lockstep:
                out = dev.policy.head(y.view(self.n, self.S, dev.dim)[:, self.S - 1, :])
            else:
                out = dev.policy.head(y, pick=self._lens[self._cur], group=self.S, pick_bias=-1)   # row len-1 of each window
        return out.view(self.n, dev.horizon, dev.action_dim)
