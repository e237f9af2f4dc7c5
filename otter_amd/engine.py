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

Both conditions hold per episode, so the engine keeps lengths and cache validity per episode: episodes that were reset
at different times sit at different window positions and still take the cached step (``otter.py:153-175`` semantics for
each of them), while the ones whose windows slide are recomputed (rows past an episode's length are zero and, by
causality, never influence earlier rows).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np
import torch

from .torch_ops import kernels as ops      # every kernel launch below is a torch.ops.otter_hip.* call
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
    """``n`` concurrent episodes.  Every episode has its own length and its own cache state:

    * an episode whose window is not full yet and whose K/V rows are valid takes the CACHED step: only its new position is
      computed, at that episode's own window position (episodes that started at different times sit at different ones);
    * an episode whose window is full (it slides from now on: every position shifts) or whose K/V rows were never produced
      (history handed over with :meth:`adopt` or written through ``OTTER.update_cache``) takes the RECOMPUTE step.

    A step in which every episode is cached at one common position, or every episode recomputes, runs over the whole
    batch with no gather; a mixed step runs the two groups one after the other on their own rows.  ``reset(ids)`` of a
    subset therefore never takes the K/V cache away from the other episodes."""

    def __init__(self, dev: OtterDevice, n_episodes: int):
        self.dev, self.n = dev, int(n_episodes)
        self.S = dev.seq_length
        d = dev.device
        self._win = [torch.zeros(self.n, self.S, dev.dim, dtype=torch.float32, device=d) for _ in range(2)]
        self._lens = [torch.zeros(self.n, dtype=torch.int32, device=d) for _ in range(2)]
        self._cur = 0
        self._cache = dev.policy.new_cache(self.n, self.S)
        self.lens_host = np.zeros(self.n, dtype=np.int64)
        self.kv_ok = np.ones(self.n, dtype=bool)       # K/V rows [0, len) of the episode are valid at their current positions
        self._graph: Optional[_EncodeGraph] = None
        self._idx_cache: dict = {}                     # episode subsets already uploaded as device index tensors
        self.last_mode = ""                            # "cached", "recompute" or "mixed": what the last step ran (tests, logs)

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
            self.kv_ok[:] = True
            return
        ids_list = [int(e) for e in episode_ids]
        if not ids_list:
            return
        if min(ids_list) < 0 or max(ids_list) >= self.n:
            raise IndexError(f"reset: episode ids must lie in [0, {self.n})")
        ids = torch.as_tensor(ids_list, dtype=torch.long, device=self.dev.device)
        self._win[self._cur].index_fill_(0, ids, 0.0)
        self._lens[self._cur].index_fill_(0, ids, 0)
        self.lens_host[ids_list] = 0
        self.kv_ok[ids_list] = True                    # an empty history is trivially valid

    def invalidate_kv(self, episode_ids: Optional[Sequence[int]] = None) -> None:
        """The windows were written by someone else (no K/V rows exist for them): recompute until the next reset."""
        if episode_ids is None:
            self.kv_ok[:] = False
        else:
            self.kv_ok[[int(e) for e in episode_ids]] = False

    def adopt(self, other: "BatchedRollout") -> None:
        """Take over the running episodes of ``other`` (same episode count, window and width): their windows and lengths are
        copied, their K/V rows -- products of other weights -- are not."""
        if (other.n, other.S, other.dev.dim) != (self.n, self.S, self.dev.dim):
            raise ValueError("adopt: the two rollouts differ in episode count, window length or width")
        self._win[self._cur].copy_(other._win[other._cur])
        self._lens[self._cur].copy_(other._lens[other._cur])
        self.lens_host[:] = other.lens_host
        self.kv_ok[:] = other.lens_host == 0

    def _index(self, ids: np.ndarray):
        """Device (int32, int64) index tensors of an episode subset; subsets repeat from step to step, so they are kept."""
        key = ids.tobytes()
        hit = self._idx_cache.get(key)
        if hit is None:
            if len(self._idx_cache) > 256:
                self._idx_cache.clear()
            t = torch.from_numpy(ids.astype(np.int32)).to(self.dev.device)
            hit = self._idx_cache[key] = (t, t.long())
        return hit

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
        before = self.lens_host
        cached = (before < self.S) & self.kv_ok                                  # per episode: only the new position is needed
        self.lens_host = np.minimum(before + 1, self.S)
        win, lens = self._win[self._cur], self._lens[self._cur]
        pol = dev.policy
        if cached.all() and (before == before[0]).all():
            # every episode at the same position: the new row of every episode, in place, no index
            self.last_mode = "cached"
            y = pol.run(win, self._cache, 1, int(before[0]))                     # (n, dim)
            out = pol.head(y)
        elif not cached.any():
            # whole windows of every episode; rows past an episode's length are zero and, by causality, touch nothing before them
            self.last_mode = "recompute"
            y = pol.run(win, self._cache, self.S, 0)                             # (n*S, dim)
            if (self.lens_host == self.S).all():
                out = pol.head(y.view(self.n, self.S, dev.dim)[:, self.S - 1, :])
            else:
                out = pol.head(y, pick=lens, group=self.S, pick_bias=-1)         # row len-1 of each window
            self.kv_ok |= before < self.S      # the rows just written stay valid until the episode's window slides
        else:
            # episodes at different positions, or some cached and some recomputing: each group on its own rows
            ids_c, ids_f = np.flatnonzero(cached), np.flatnonzero(~cached)
            self.last_mode = "mixed" if ids_f.size else "cached"
            out = torch.empty(self.n, dev.horizon * dev.action_dim, dtype=torch.float32, device=dev.device)
            i32, i64 = self._index(ids_c)
            out.index_copy_(0, i64, pol.head(pol.run_rows(win, self._cache, i32, lens)))
            if ids_f.size:
                i32, i64 = self._index(ids_f)
                win_f = win.index_select(0, i64)
                lens_f = lens.index_select(0, i64)
                y = pol.run(win_f, pol.new_cache(ids_f.size, self.S), self.S, 0)     # a scratch cache: these rows are not kept
                out.index_copy_(0, i64, pol.head(y, pick=lens_f, group=self.S, pick_bias=-1))
        return out.view(self.n, dev.horizon, dev.action_dim)
