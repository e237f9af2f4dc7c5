"""Training step of the trainable modules on the MI355X: forward with saved activations + hand-written backward.

In the reference every parameter outside CLIP is trained (``otter/policy/otter.py:75-128``; 31,055,290 values) with
torch autograd under bf16 autocast (``otter/util/engine.py:44-66``).  Here the frozen towers run exactly as in
inference (16-bit MFMA kernels, no gradient), and everything downstream runs in fp32 through the kernels of
``csrc/train_ops.hip``.  Differentiation is a small reverse-mode tape: every op below enqueues its forward kernels,
records a closure that enqueues the backward kernels, and :meth:`Tape.backward` replays the closures in reverse.
Gradient kernels accumulate (``+=``) into zero-initialised buffers, so a tensor consumed twice needs no add kernel,
and a residual connection hands its output-gradient buffer on to the skip input without a copy.

``TailLoss`` wraps the whole thing in one ``torch.autograd.Function`` whose inputs are the trainable parameters, so
``loss = model(images, text, proprio, gt_actions); loss.backward()`` works as it does with the reference model
(gradient accumulation, ``DistributedDataParallel`` hooks and ``torch.amp.GradScaler`` included).
"""
from __future__ import annotations

import threading
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor


class _GradArena:
    """Activation-gradient buffers of one forward + backward pass, carved out of ONE zero-filled allocation (about 150 separate
    zero fills per pass otherwise, each a launch).  The size is learnt from the previous pass; whatever does not fit falls
    back to its own allocation."""

    def __init__(self, device, hint: int):
        self.buf = torch.zeros(hint, dtype=torch.float32, device=device) if hint > 0 else None
        self.off = 0          # elements handed out of ``buf``
        self.used = 0         # elements asked for in all (the next pass's hint)

    def take(self, like: Tensor) -> Tensor:
        n = like.numel()
        n_al = (n + 63) // 64 * 64
        self.used += n_al
        if self.buf is not None and like.device == self.buf.device and self.off + n_al <= self.buf.numel():
            out = self.buf[self.off:self.off + n].view(like.shape)
            self.off += n_al
            return out
        return torch.zeros_like(like, memory_format=torch.contiguous_format)


_arena_tls = threading.local()


class Var:
    """A device tensor (2-D row-major view unless noted) and, lazily, its gradient buffer."""

    __slots__ = ("t", "g", "needs_grad")

    def __init__(self, t: Tensor, needs_grad: bool = True, g: Optional[Tensor] = None):
        self.t, self.g, self.needs_grad = t, g, needs_grad

    def grad(self) -> Tensor:
        if self.g is None:
            arena = getattr(_arena_tls, "arena", None)
            if arena is not None and self.t.dtype == torch.float32:
                self.g = arena.take(self.t)
            else:
                self.g = torch.zeros_like(self.t, memory_format=torch.contiguous_format)
        return self.g


class Tape:
    def __init__(self):
        self._steps: List[Callable[[], None]] = []

    def record(self, fn: Callable[[], None]) -> None:
        self._steps.append(fn)

    def __len__(self) -> int:
        return len(self._steps)

    def backward(self, on_step: Optional[Callable[[int], None]] = None) -> None:
        """Replay the closures last to first; ``on_step(i)`` runs after closure i (every closure recorded at or after
        position i has then run: the gradients of whatever was first used at i or later are final)."""
        for i in range(len(self._steps) - 1, -1, -1):
            self._steps[i]()
            if on_step is not None:
                on_step(i)
        self._steps.clear()


# ---------------------------------------------------------------------------------------------------------- tape ops
def linear(tape: Tape, x: Var, w: Var, b: Optional[Var], out: Optional[Var] = None, residual: Optional[Var] = None) -> Var:
    """y = x W^T + b (+ residual); ``out`` lets the caller place y (and its gradient) inside a larger buffer."""
    m, n = x.t.shape[0], w.t.shape[0]
    if out is None:
        out = Var(torch.empty(m, n, dtype=torch.float32, device=x.t.device))
    if residual is not None:
        out.t.copy_(residual.t)
    ops.sgemm(x.t, w.t, out.t, trans_b=True, bias=None if b is None else b.t, accumulate=residual is not None)

    def bwd():
        if out.g is None:
            return
        dy = out.g
        db = b is not None and b.needs_grad
        if w.needs_grad:                                   # the bias gradient (column sums of dy) rides on the same launch
            ops.sgemm(dy, x.t, w.grad(), trans_a=True, accumulate=True, a_rowsum=b.grad() if db else None)
        elif db:
            ops.col_sum(dy, b.grad(), accumulate=True)
        if x.needs_grad:
            ops.sgemm(dy, w.t, x.grad(), accumulate=True)
        if residual is not None and residual.needs_grad:
            if residual.g is None and dy.is_contiguous():
                residual.g = dy                       # hand the buffer on: later (earlier-in-forward) ops accumulate into it
            else:
                ops.add_inplace(residual.grad(), dy.contiguous())
    tape.record(bwd)
    return out


def activation(tape: Tape, z: Var, act: int) -> Var:
    y = Var(ops.act_fwd(z.t, act))

    def bwd():
        if y.g is not None and z.needs_grad:
            ops.act_bwd(z.t, y.g, z.grad(), act)
    tape.record(bwd)
    return y


def layernorm(tape: Tape, x: Var, gamma: Var, beta: Var, out: Optional[Var] = None) -> Var:
    y = ops.layernorm(x.t, gamma.t, beta.t, out_dtype=torch.float32, out=None if out is None else out.t)
    out = out if out is not None else Var(y)

    def bwd():
        if out.g is None:
            return
        dgamma = gamma.grad() if gamma.needs_grad else None
        dbeta = beta.grad() if beta.needs_grad else None
        if (dgamma is None) != (dbeta is None):       # the kernel produces both or neither: the frozen one gets a throw-away buffer
            dgamma = dgamma if dgamma is not None else torch.zeros_like(gamma.t)
            dbeta = dbeta if dbeta is not None else torch.zeros_like(beta.t)
        if not x.needs_grad and dgamma is None:
            return
        ops.layernorm_bwd(x.t, gamma.t, out.g, x.grad() if x.needs_grad else None, dgamma, dbeta)
    tape.record(bwd)
    return out


def attention(tape: Tape, q: Var, k: Var, v: Var, batch: int, lq: int, lk: int, heads: int, causal: bool,
              key_mask: Optional[Tensor] = None, broadcast_q: bool = False) -> Var:
    """q rows (batch*lq or lq when broadcast, d), k / v rows (batch*lk, d) -- views into wider buffers allowed."""
    d = k.t.shape[1]

    def as3(t: Tensor, l: int, nb: int) -> Tensor:
        return t.as_strided((nb, l, d), (l * t.stride(0), t.stride(0), 1), t.storage_offset())

    q3 = as3(q.t, lq, 1 if broadcast_q else batch)
    k3, v3 = as3(k.t, lk, batch), as3(v.t, lk, batch)
    o = ops.small_attn_fwd(q3, k3, v3, heads, causal, key_mask, broadcast_q)
    out = Var(o.view(batch * lq, d))

    def bwd():
        if out.g is None:
            return
        dq_full = None
        if q.needs_grad:
            dq_full = torch.zeros(batch, lq, d, dtype=torch.float32, device=o.device) if broadcast_q else as3(q.grad(), lq, batch)
        dk = as3(k.grad(), lk, batch) if k.needs_grad else None
        dv = as3(v.grad(), lk, batch) if v.needs_grad else None
        ops.small_attn_bwd(q3, k3, v3, out.g.view(batch, lq, d), dq_full, dk, dv, heads, causal, key_mask, broadcast_q)
        if broadcast_q and q.needs_grad:              # the shared queries collect the gradient of every batch element
            ops.col_sum(dq_full.view(batch, lq * d), q.grad().view(-1), accumulate=True)
    tape.record(bwd)
    return out


def tile_rows(tape: Tape, x: Var, times: int) -> Var:
    """(R, d) -> (times*R, d), row b*R + r = x[r] (``query.expand(B, -1, -1)``, transformer.py:223)."""
    r, d = x.t.shape
    out = Var(x.t.unsqueeze(0).expand(times, r, d).reshape(times * r, d))

    def bwd():
        if out.g is not None and x.needs_grad:
            ops.col_sum(out.g.view(times, r * d), x.grad().view(-1), accumulate=True)
    tape.record(bwd)
    return out


# ------------------------------------------------------------------------------------------------- the trainable tail
class TrainableTail:
    """Forward / backward of everything downstream of the CLIP towers for a (B, T) batch."""

    def __init__(self, model):
        self.model = model
        # every parameter outside the frozen towers, trainable or not: a frozen one (action-head-only fine-tuning, a model
        # loaded by OtterInference) takes part in the forward pass and simply gets no gradient buffer
        self.params: Dict[str, torch.nn.Parameter] = {n: p for n, p in model.named_parameters() if not n.startswith("clip_model.")}
        self.buffers = dict(model.named_buffers())
        self._tape: Optional[Tape] = None
        self.first_use: Dict[str, int] = {}       # parameter name -> tape position at its first use in the running forward pass

    def trainable_names(self) -> List[str]:
        return [n for n, p in self.params.items() if p.requires_grad]

    def _dims(self, device, clip_dtype):
        """Sizes and constant tables (cached): pooled widths, transposed pos_emb tables, the window positional table."""
        # the packed tables follow their source buffers: in-place writes bump the version, re-pointing changes the address
        packed = [b for n, b in self.buffers.items() if n == "pe.pe" or n.endswith(".pos_emb")]
        key = (device, clip_dtype, tuple((b._version, b.data_ptr()) for b in packed))
        if getattr(self, "_dims_key", None) != key:
            from types import SimpleNamespace
            m = self.model
            cams = len(m.camera_keys)
            q_t, q_v = self.params["text_pooling.query"], self.params["vision_poolings.0.query"]
            d_text, d_vis = q_t.shape[1] * q_t.shape[2], q_v.shape[1] * q_v.shape[2]
            d_prop = self.params["proprio_encoder.mlp.4.weight"].shape[0]
            pos_t = torch.stack([ops.pos_emb_pack(self.buffers[f"visual_extraction.{c}.pos_emb"].detach().float().to(device).contiguous(),
                                                       clip_dtype) for c in range(cams)]).contiguous()
            pe = self.buffers["pe.pe"].detach()[0].float().to(device).contiguous()
            self._dims_cache = SimpleNamespace(d_text=d_text, d_vis=d_vis, d_prop=d_prop, f_t=d_text + cams * d_vis + d_prop, pos_t=pos_t, pe=pe)
            self._dims_key = key
        return self._dims_cache

    # -- parameter access ------------------------------------------------------------------------------------------
    def _var(self, grads: Dict[str, Tensor], name: str, shape2d: Optional[Tuple[int, int]] = None) -> Var:
        """The parameter as a tape variable; it collects a gradient iff ``grads`` holds a buffer for it."""
        if self._tape is not None:
            self.first_use.setdefault(name, len(self._tape))
        t = self.params[name].detach()
        g = grads.get(name)
        if shape2d is not None:
            t = t.view(*shape2d)
            g = None if g is None else g.view(*shape2d)
        return Var(t, g is not None, g)

    def _slice(self, v: Var, lo: int, hi: int) -> Var:
        return Var(v.t[lo:hi], v.needs_grad, None if v.g is None else v.g[lo:hi])

    # -- modules ------------------------------------------------------------------------------------------------------
    def _pooling(self, tape: Tape, G, prefix: str, x: Var, n_obs: int, nk: int, heads: int, mask: Optional[Tensor], out: Var) -> None:
        """AttentionPooling (transformer.py:162-232) on x (n_obs*nk, D_in) -> out (n_obs, readouts*d) view."""
        query = self._var(G, prefix + "query")
        _, r, d = query.t.shape
        q = Var(query.t.view(r, d), query.needs_grad, None if query.g is None else query.g.view(r, d))   # shared by every observation until the first residual
        shared = True
        i = 0
        while (prefix + f"blocks.{i}.q_layer_norm.weight") in self.params:
            p = prefix + f"blocks.{i}."
            P = lambda s: self._var(G, p + s)                        # noqa: E731
            qn = layernorm(tape, q, P("q_layer_norm.weight"), P("q_layer_norm.bias"))
            kvn = layernorm(tape, x, P("kv_layer_norm.weight"), P("kv_layer_norm.bias"))
            bias = P("attention.in_proj_bias")
            qq = linear(tape, qn, P("attention.q_proj_weight"), self._slice(bias, 0, d))
            kv = Var(torch.empty(n_obs * nk, 2 * d, dtype=torch.float32, device=x.t.device))
            kv.grad()
            kk = linear(tape, kvn, P("attention.k_proj_weight"), self._slice(bias, d, 2 * d), out=Var(kv.t[:, :d], True, kv.g[:, :d]))
            vv = linear(tape, kvn, P("attention.v_proj_weight"), self._slice(bias, 2 * d, 3 * d), out=Var(kv.t[:, d:], True, kv.g[:, d:]))
            ctx = attention(tape, qq, kk, vv, n_obs, r, nk, heads, False, mask, broadcast_q=shared)
            res = tile_rows(tape, qn, n_obs) if shared else qn          # residual on the NORMALISED query (transformer.py:189-192)
            shared = False
            q1 = linear(tape, ctx, P("attention.out_proj.weight"), P("attention.out_proj.bias"), residual=res)
            h = layernorm(tape, q1, P("layer_norm.weight"), P("layer_norm.bias"))
            z = linear(tape, h, P("mlp.fc1.weight"), P("mlp.fc1.bias"))
            a = activation(tape, z, ops.ACT_GELU)
            q = linear(tape, a, P("mlp.fc2.weight"), P("mlp.fc2.bias"), residual=q1)
            i += 1
        # final LN writes (n_obs*r, d) rows straight into the (n_obs, r*d) slice of the concatenated feature buffer
        if out.t.stride(0) == r * d:
            layernorm(tape, q, self._var(G, prefix + "layer_norm.weight"), self._var(G, prefix + "layer_norm.bias"),
                      out=Var(out.t.view(n_obs * r, d), True, out.g.view(n_obs * r, d)))
        else:
            y = layernorm(tape, q, self._var(G, prefix + "layer_norm.weight"), self._var(G, prefix + "layer_norm.bias"))
            out.t.copy_(y.t.view(n_obs, r * d))

            def bwd():
                y.grad().view(n_obs, r * d).copy_(out.g)
            tape.record(bwd)
            # the copy-back closure must run BEFORE the LN backward, i.e. be recorded after it: it is.

    def forward_backward(self, images: Tensor, tokens_host: Tensor, proprio: Tensor, gt_actions: Tensor, text_mask: Optional[Tensor],
                         grads: Optional[Dict[str, Tensor]], grad_scale: float = 1.0, exchange=None) -> Tensor:
        """images (B, T, cams, 3, R, R) uint8 / fp32 on the device; tokens (B, ctx) int32 on the host; proprio (B, T, P_in);
        gt_actions (B, T, horizon, action_dim).  With ``grads`` (name -> zero-initialised fp32 buffer) the gradients of
        ``grad_scale * loss`` are accumulated into it.  ``exchange`` (a :class:`otter_amd.trainer.BucketedExchange`) is told,
        while the backward pass runs, which parameters' gradients are final, so that the ranks' gradient exchange overlaps
        the rest of the pass (what DistributedDataParallel's bucket hooks do, ``script/train.py:116-119``).
        Returns the loss as a (1,) device tensor."""
        if grads is None:
            return self._forward_backward(images, tokens_host, proprio, gt_actions, text_mask, grads, grad_scale)
        arena = _GradArena(images.device, getattr(self, "_arena_hint", 0))
        _arena_tls.arena = arena
        try:
            return self._forward_backward(images, tokens_host, proprio, gt_actions, text_mask, grads, grad_scale, exchange)
        finally:
            _arena_tls.arena = None
            self._tape = None
            self._arena_hint = arena.used

    def _forward_backward(self, images: Tensor, tokens_host: Tensor, proprio: Tensor, gt_actions: Tensor, text_mask: Optional[Tensor],
                          grads: Optional[Dict[str, Tensor]], grad_scale: float = 1.0, exchange=None) -> Tensor:
        model = self.model
        clip = model.clip_device()
        dm = self._dims(clip.device, clip.dtype)
        b, t, cams = images.shape[:3]
        m = b * t
        k = model.first_k_tokens
        want = grads is not None
        G = grads if want else {}                                                   # no buffers: nothing collects a gradient
        tape = Tape()
        self._tape, self.first_use = tape, {}
        device = clip.device

        # ---- frozen towers (no gradient) ----
        flat = images.reshape(m * cams, *images.shape[3:])
        flat = flat if flat.dtype == torch.uint8 else flat.float()
        tokens, p, d = clip.vision_tokens(flat.contiguous())
        tf_raw = clip.text_tokens_cached(tokens_host)                        # (B, K, D) fp32, un-normalised
        temperature = torch.stack([self.params[f"visual_extraction.{c}.temperature"].detach().float() for c in range(cams)]).contiguous()
        ta = ops.text_aware_pool(tokens, tf_raw, dm.pos_t, temperature, m, cams, t, n_patches=p)    # (cams, M, K, D)
        ta_g = torch.zeros_like(ta) if want else None

        f_t = dm.f_t
        combined = Var(torch.zeros(m, f_t, dtype=torch.float32, device=device))
        combined.grad()
        mask_m = None
        if model.pool_true_text:
            if text_mask is None:
                raise ValueError("pool_true_text models need a text mask")
            mask_m = text_mask.to(device=device, dtype=torch.uint8).repeat_interleave(t, dim=0).contiguous()
        heads = model.pooling_heads
        off = dm.d_text
        for c in range(cams):
            x = Var(ta[c].view(m * k, d), want, None if not want else ta_g[c].view(m * k, d))
            self._pooling(tape, G, f"vision_poolings.{c}.", x, m, k, heads, mask_m,
                          Var(combined.t[:, off:off + dm.d_vis], True, combined.g[:, off:off + dm.d_vis]))
            off += dm.d_vis
        tfn = ops.l2norm_rows(tf_raw.view(b * k, d))
        tf_rep = torch.empty(m, k * d, dtype=torch.float32, device=device)
        ops.repeat_rows(tfn.view(b, k * d), tf_rep, t)                              # repeat_interleave(T) of otter.py:200
        self._pooling(tape, G, "text_pooling.", Var(tf_rep.view(m * k, d), False), m, k, heads, mask_m,
                      Var(combined.t[:, : dm.d_text], True, combined.g[:, : dm.d_text]))
        # proprio encoder (models.py:38-51)
        h = Var(proprio.reshape(m, -1).to(device, torch.float32).contiguous(), False)
        for i, last in ((0, False), (2, False), (4, True)):
            w, bb = self._var(G, f"proprio_encoder.mlp.{i}.weight"), self._var(G, f"proprio_encoder.mlp.{i}.bias")
            if last:
                linear(tape, h, w, bb, out=Var(combined.t[:, off:off + dm.d_prop], True, combined.g[:, off:off + dm.d_prop]))
            else:
                h = activation(tape, linear(tape, h, w, bb), ops.ACT_RELU)
        x = linear(tape, combined, self._var(G, "input_projection.weight"), self._var(G, "input_projection.bias"))
        # + positional embedding by window position (transformer.py:23-24): gradient passes through unchanged
        xp = Var(ops.add_pe(x.t.view(b, t, -1), dm.pe, t, 0))

        def pe_bwd(x_in=x, x_out=xp):
            if x_out.g is not None:
                x_in.g = x_out.g if x_in.g is None else ops.add_inplace(x_in.g, x_out.g)
        tape.record(pe_bwd)
        x = xp
        dim = x.t.shape[1]
        nh = model.transformer_heads
        i = 0
        while f"policy.layers.{i}.pre_ln.weight" in self.params:
            pfx = f"policy.layers.{i}."
            P = lambda s: self._var(G, pfx + s)                          # noqa: E731
            hn = layernorm(tape, x, P("pre_ln.weight"), P("pre_ln.bias"))
            qkv = Var(torch.empty(m, 3 * dim, dtype=torch.float32, device=device))
            qkv.grad()
            parts = []
            for j, nm in enumerate(("query", "key", "value")):
                sl = slice(j * dim, (j + 1) * dim)
                parts.append(linear(tape, hn, P(f"attention.{nm}.weight"), P(f"attention.{nm}.bias"),
                                    out=Var(qkv.t[:, sl], True, qkv.g[:, sl])))
            ctx = attention(tape, parts[0], parts[1], parts[2], b, t, t, nh, True)
            x = linear(tape, ctx, P("attention.out.weight"), P("attention.out.bias"), residual=x)
            hn = layernorm(tape, x, P("post_ln.weight"), P("post_ln.bias"))
            a = activation(tape, linear(tape, hn, P("mlp.0.weight"), P("mlp.0.bias")), ops.ACT_GELU)
            x = linear(tape, a, P("mlp.2.weight"), P("mlp.2.bias"), residual=x)
            i += 1
        hh = activation(tape, linear(tape, x, self._var(G, "action_head.mlp.0.weight"), self._var(G, "action_head.mlp.0.bias")), ops.ACT_RELU)
        pred = linear(tape, hh, self._var(G, "action_head.mlp.2.weight"), self._var(G, "action_head.mlp.2.bias"))
        gt = gt_actions.to(device, torch.float32).reshape(m, -1).contiguous()
        loss, dpred = ops.loss_fwd_bwd(pred.t, gt, model.use_mse, grad_scale, want_grad=want)
        if not want:
            self._tape = None
            return loss
        pred.g = dpred
        if exchange is not None:
            # the temperatures get their gradient after the tape (below): their buckets go out last
            exchange.begin({n: i for n, i in self.first_use.items() if not n.endswith(".temperature")})
        tape.backward(None if exchange is None else exchange.after_step)
        # temperature: the only trainable input of the fused extraction kernel
        pos = torch.stack([self.buffers[f"visual_extraction.{c}.pos_emb"].detach().float() for c in range(cams)]).contiguous()
        part = ops.text_aware_pool_temperature_grad(tokens, tf_raw, pos, temperature, ta_g, m, cams, t, n_patches=p)
        tg = torch.zeros(cams, dtype=torch.float32, device=device)
        ops.col_sum(part, tg, accumulate=False)
        for c in range(cams):
            if f"visual_extraction.{c}.temperature" in grads:
                ops.add_inplace(grads[f"visual_extraction.{c}.temperature"].view(1), tg[c:c + 1])
        if exchange is not None:
            exchange.finish()
        return loss


class TailLoss(torch.autograd.Function):
    """loss = OTTER(images, text, proprio, gt_actions) as a differentiable function of the trainable parameters."""

    @staticmethod
    def forward(ctx, tail: TrainableTail, images, tokens_host, proprio, gt_actions, text_mask, names, *params):
        grads = {n: torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format) for n, p in zip(names, params)}
        loss = tail.forward_backward(images, tokens_host, proprio, gt_actions, text_mask, grads, 1.0)
        ctx.grads = [grads[n] for n in names]
        return loss.view(())

    @staticmethod
    def backward(ctx, grad_loss):
        # d(loss)/d(param) were produced with the forward pass; the incoming factor (loss scaling, accumulation
        # weights) multiplies them
        s = float(grad_loss.item())
        out = []
        for g in ctx.grads:
            if s != 1.0:
                ops.scale_inplace(g.view(-1), s)
            out.append(g)
        return (None, None, None, None, None, None, None, *out)
