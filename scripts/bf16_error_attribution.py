#!/usr/bin/env python3
"""Which tensors of the CLIP towers spend the action-error budget when they are rounded to bf16?

The path's tolerance is 1e-3 on actions against the fp32 reference.  fp16 tower operands meet it, bf16 operands do not
(measured 1.2e-3 to 4e-3).  This diagnostic runs the default ViT-L/14 configuration with fp16 operands and an fp32
residual stream and injects bf16 rounding into ONE class of tensors at a time (and into one quarter of the vision
tower's depth at a time), then reports the largest action error against the fp32 CPU oracle for each setting.

A bf16 value inside fp16's normal range is exactly representable in fp16, so `x.to(bfloat16).to(float16)` followed by the
fp16 kernels is the bf16 kernel's arithmetic on that tensor (fp32 accumulation either way); attention is run in the real
bf16 kernel (its probabilities are rounded inside the kernel).  A measuring tool: it uses torch casts between kernels and
is not part of the product path.

    python scripts/bf16_error_attribution.py [--episodes 6] [--steps 2] > profiles/r03_bf16_error_attribution.txt
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OTTER_HIP_GRAPH"] = "0"                     # eager launches: the casts between kernels differ from setting to setting

from oracle import otter_ref as R                      # noqa: E402  (the checker)
from otter_amd import ops                              # noqa: E402
from otter_amd.config import ModelConfig, SharedConfig  # noqa: E402
from otter_amd.device_model import CLIP_MEAN, CLIP_STD, ClipDevice, OtterDevice  # noqa: E402
from otter_amd.engine import BatchedRollout            # noqa: E402
from otter_amd.otter import OTTER                      # noqa: E402
from otter_amd.params import otter_specs, random_state_dict  # noqa: E402

F16, BF16 = torch.float16, torch.bfloat16
CLASSES = ("weights", "ln_out", "attention", "attn_out", "mlp_hidden", "embed", "head", "pool")


def rb(t: torch.Tensor) -> torch.Tensor:
    """bf16 rounding of a 16-bit / fp32 tensor, returned in fp16 (exact for |x| in fp16's normal range)."""
    return t.to(BF16).to(F16)


class SweepClip(ClipDevice):
    """fp16 towers with an fp32 residual stream; `self.on` names the tensor classes that carry bf16 rounding, `self.layers`
    the vision blocks (and `self.text` whether the text tower) they apply to."""

    def __init__(self, sd, arch, first_k, device):
        super().__init__(sd, arch, first_k, device, F16, stream_dtype=torch.float32)
        self.on, self.layers, self.text = set(), range(0), False
        # a second copy of every block weight, rounded to bf16 first
        for blocks in (self.vblocks, self.tblocks):
            for b in blocks:
                for name in ("wqkv", "wo", "w_fc", "w_proj"):
                    setattr(b, name + "_r", rb(getattr(b, name)))
        self.w_conv_r, self.w_vproj_r, self.w_tproj_r = rb(self.w_conv), rb(self.w_vproj), rb(self.w_tproj)

    def _active(self, i: int, text: bool) -> bool:
        return self.text if text else (i in self.layers)

    def _run_blocks(self, x, blocks, n_seq, seq_len, heads, causal, stop_after_last_attention):
        text = blocks is self.tblocks
        last = None
        for i, b in enumerate(blocks):
            act = self._active(i, text)
            on = self.on if act else ()
            w = (lambda n: getattr(b, n + "_r")) if "weights" in on else (lambda n: getattr(b, n))
            h = ops.layernorm(x, *b.ln1, out_dtype=F16)
            if "ln_out" in on:
                h = rb(h)
            qkv = ops.gemm(h, w("wqkv"), b.n_qkv, b.bqkv)
            if "attention" in on:                                    # q, k, v and the probabilities in bf16: the real kernel
                a = ops.attn_d64(qkv.to(BF16), n_seq, seq_len, heads, causal).to(F16)
            else:
                a = ops.attn_d64(qkv, n_seq, seq_len, heads, causal)
                if "attn_out" in on:
                    a = rb(a)
            if stop_after_last_attention and i == len(blocks) - 1:
                last = ops.gemm(a, w("wo"), b.width, b.bo, out_f32=True)
                break
            ops.gemm(a, w("wo"), b.width, b.bo, residual=x, out=x)
            h = ops.layernorm(x, *b.ln2, out_dtype=F16)
            if "ln_out" in on:
                h = rb(h)
            m = ops.gemm(h, w("w_fc"), b.n_fc, b.b_fc, act=self.act)
            if "mlp_hidden" in on:
                m = rb(m)
            ops.gemm(m, w("w_proj"), b.width, b.b_proj, residual=x, out=x)
        return last if stop_after_last_attention else x

    def vision_tokens(self, images):
        n, _, res, _ = images.shape
        p = (res // self.patch) ** 2
        emb, head = "embed" in self.on and len(self.layers) > 0, "head" in self.on and len(self.layers) > 0
        cols = ops.im2col_patches(images, self.patch, self.k_im2col, CLIP_MEAN, CLIP_STD, F16)
        tok = ops.gemm(rb(cols) if emb else cols, self.w_conv_r if emb else self.w_conv, self.vw, None, out_f32=True)
        x = ops.vit_embed_lnpre(tok, self.cls, self.vpos, self.ln_pre[0], self.ln_pre[1], n, p, out_dtype=torch.float32)
        last = self._run_blocks(x, self.vblocks, n, p + 1, self.v_heads, False, True)
        h = ops.layernorm(last, *self.ln_post, out_dtype=F16)
        feats = ops.gemm(rb(h) if head else h, self.w_vproj_r if head else self.w_vproj, self.embed, None)
        if head:
            feats = rb(feats)
        return feats.view(n, p + 1, self.embed), p, self.embed

    def text_tokens(self, ids, pooled_row=False):
        b, k = ids.shape[0], self.first_k
        x = ops.text_embed(ids, k, self.tok_emb, self.tpos, out_dtype=torch.float32)
        x = self._run_blocks(x, self.tblocks, b, k, self.t_heads, True, False)
        h = ops.layernorm(x, *self.ln_final, out_dtype=F16)
        head = self.text and "head" in self.on
        return ops.gemm(rb(h) if head else h, self.w_tproj_r if head else self.w_tproj, self.embed, None, out_f32=True).view(b, k, self.embed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=6)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    device = torch.device("cuda:0")
    mc, sc = ModelConfig(), SharedConfig()
    sd = random_state_dict(otter_specs(mc, sc), seed=0)
    model = OTTER(mc, sc)
    clip = SweepClip({k: v for k, v in sd.items() if k.startswith("clip_model.")}, model.clip_arch, mc.first_k_tokens, device)
    dev = OtterDevice(sd, model.clip_arch, first_k=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                      num_cameras=len(sc.camera_keys), seq_length=sc.seq_length, action_horizon=sc.action_horizon, action_dim=mc.action_dim,
                      pool_true_text=mc.pool_true_text, device=device, clip_dtype=F16, clip=clip)
    pos_bf16 = torch.stack([ops.pos_emb_pack(sd[f"visual_extraction.{c}.pos_emb"].float().to(device).contiguous(), BF16) for c in range(2)]).contiguous()
    real_pool = ops.text_aware_pool

    def pool(patches, text, pos_packed, temperature, n_obs, cams, t_rep, n_patches=None):
        if "pool" in clip.on and len(clip.layers) > 0:                # the pooling kernel's own bf16 instance
            return real_pool(patches.to(BF16), text, pos_bf16, temperature, n_obs, cams, t_rep, n_patches=n_patches)
        return real_pool(patches, text, pos_packed, temperature, n_obs, cams, t_rep, n_patches=n_patches)

    ops.text_aware_pool = pool
    n, steps = args.episodes, args.steps
    rng = np.random.default_rng(41)
    g = torch.Generator().manual_seed(41)
    images = torch.randint(0, 256, (steps, n, 2, 3, 224, 224), dtype=torch.uint8, generator=g)
    proprio = torch.from_numpy(rng.standard_normal((steps, n, 10)).astype(np.float32))
    tokens = torch.zeros(n, 77, dtype=torch.int32)
    for e in range(n):
        ln = 3 + e % 5
        tokens[e, 0], tokens[e, 1 + ln] = 49406, 49407
        tokens[e, 1:1 + ln] = torch.from_numpy(rng.integers(1, 49406, ln).astype(np.int32))
    t0 = time.time()
    ref = torch.zeros(steps, n, sc.action_horizon, mc.action_dim)
    for e in range(n):
        ro = R.Rollout(sd, R.Dims())
        for t in range(steps):
            ref[t, e] = ro.step(R.clip_transform(images[t, e]), tokens[e], proprio[t, e])
    print(f"# fp32 CPU oracle: {n} episodes x {steps} steps in {time.time() - t0:.1f} s; |action| max {ref.abs().max():.3f}, rms {ref.pow(2).mean().sqrt():.3f}")

    def measure(on, layers, text):
        clip.on, clip.layers, clip.text = set(on), layers, text
        clip._text_cache.clear()
        tf = clip.text_tokens_cached(tokens)
        ro = BatchedRollout(dev, n)
        worst = rms = 0.0
        for t in range(steps):
            got = ro.step(images[t].view(n * 2, 3, 224, 224).to(device), tf, proprio[t].to(device)).cpu()
            d = got - ref[t]
            worst, rms = max(worst, d.abs().max().item()), rms + d.pow(2).mean().item()
        return worst, (rms / steps) ** 0.5

    all_layers = range(len(clip.vblocks))
    rows = [("fp16 operands, fp32 stream (no bf16 anywhere)", (), range(0), False)]
    rows += [(f"bf16: {c} only (both towers, all blocks)", (c,), all_layers, True) for c in CLASSES]
    rows += [("bf16: every class, vision tower only", CLASSES, all_layers, False),
             ("bf16: every class, text tower only", CLASSES, range(0), True)]
    q = len(clip.vblocks) // 4
    rows += [(f"bf16: every class, vision blocks {i * q}-{(i + 1) * q - 1} only", CLASSES, range(i * q, (i + 1) * q), False) for i in range(4)]
    rows += [("bf16: everything except the weights", tuple(c for c in CLASSES if c != "weights"), all_layers, True),
             ("bf16: everything except attention", tuple(c for c in CLASSES if c != "attention"), all_layers, True),
             ("bf16: everything except ln_out", tuple(c for c in CLASSES if c != "ln_out"), all_layers, True),
             ("bf16: everything except mlp_hidden", tuple(c for c in CLASSES if c != "mlp_hidden"), all_layers, True),
             ("bf16: weights + attention only", ("weights", "attention"), all_layers, True),
             ("bf16: every class, both towers", CLASSES, all_layers, True)]
    print(f"# classes: weights = every tower matrix; ln_out = LayerNorm outputs feeding in_proj / c_fc; attention = q|k|v, probabilities and "
          f"output through the bf16 attention kernel; attn_out = the attention output alone; mlp_hidden = c_fc output after the activation; "
          f"embed = patch pixels + conv1; head = ln_post / ln_final output, projection matrices, projected tokens; pool = pooling kernel in bf16")
    print(f"{'setting':64s} {'max |action err|':>18s} {'rms':>10s}   (tolerance 1e-3)")
    for name, on, layers, text in rows:
        worst, rms = measure(on, layers, text)
        print(f"{name:64s} {worst:18.2e} {rms:10.2e}   {'ok' if worst <= 1e-3 else 'FAILS'}")
    # the product's two configurations through the same inputs, for reference
    ops.text_aware_pool = real_pool
    for dt, tdt, label in ((F16, None, "product path: fp16 operands, fp16 stream, folded LayerNorm"),
                           (BF16, None, "product path: bf16 vision tower + fp16 text tower, fp32 stream"),
                           (BF16, BF16, "product path: bf16 operands in both towers, fp32 stream")):
        m = OTTER(mc, sc)
        m.load_state_dict(sd, strict=True)
        m.clip_compute_dtype, m.clip_text_dtype = dt, tdt
        m = m.to(device).eval()
        d = m.device_model()
        tf = d.clip.text_tokens_cached(tokens)
        ro = BatchedRollout(d, n)
        worst = 0.0
        for t in range(steps):
            got = ro.step(images[t].view(n * 2, 3, 224, 224).to(device), tf, proprio[t].to(device)).cpu()
            worst = max(worst, (got - ref[t]).abs().max().item())
        print(f"{label:64s} {worst:18.2e} {'':10s}   {'ok' if worst <= 1e-3 else 'FAILS'}")
        del m, d, ro


if __name__ == "__main__":
    main()
