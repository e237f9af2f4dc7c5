"""Device-side execution of the OTTER hot path: packed weights + kernel sequencing.

Everything here enqueues HIP kernels from ``libotter_hip.so`` through :mod:`otter_amd.ops`; torch only
supplies device memory and the stream.  The arithmetic follows the reference's module tree
(``otter/policy/otter.py:177-337``), restructured for the device:

* CLIP towers: every GEMM / attention operand is a 16-bit type (bf16 or fp16, fp32 accumulation), LayerNorm feeds
  the next GEMM directly in that type, bias / activation / residual-add are GEMM epilogues.  The residual stream
  stays in HBM: in fp16 with fp16 towers (what the reference's own GPU path carries -- ``clip.load`` converts the
  model to fp16 -- at half the bytes of the largest memory stream of the step), in fp32 with bf16 towers (8
  mantissa bits are too few for a stream that accumulates 47 updates);
* everything after the towers (pooling projections, input projection, policy, action head) is <0.5 % of the
  FLOPs and runs on the fp32 MFMA path so that actions track an fp32 reference to ~1e-4;
* the last vision block stops after its attention out-projection (the reference hooks that activation and
  discards the rest, ``otter.py:141-143,190``); the text tower computes only the first ``first_k_tokens``
  positions (exact under the causal mask) and is cached per prompt;
* patch / text L2-normalisation, similarity, temperature softmax and the weighted patch sum are one fused kernel;
* the policy keeps a GPU-resident window of inputs and a per-layer K/V history.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch

from .torch_ops import kernels as ops      # every kernel launch below is a torch.ops.otter_hip.* call
from .config import ClipArch
from .pooling import PoolingWeights

Tensor = torch.Tensor

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # otter/policy/vision_tf.py:4,10
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _f32(t: Tensor, device) -> Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


# OTTER_FUSE_STATS=0: the residual products and the row statistics behind them as separate launches (same results; A/B switch)
_FUSE_STATS = os.environ.get("OTTER_FUSE_STATS", "1") != "0"


class _Block:
    """One pre-LN residual attention block with packed weights (CLIP resblock or policy layer)."""

    def __init__(self, device, dtype, ln1, wqkv, bqkv, wo, bo, ln2, w_fc, b_fc, w_proj, b_proj, fold_ln: bool = False):
        """``fold_ln``: ln_1 / ln_2 are folded into in_proj / c_fc (``ops.fold_layernorm``): those two GEMMs then read the
        residual stream directly and the normalised activations are never written."""
        self.dtype = dtype
        self.fold_ln = fold_ln
        self.ln1 = (_f32(ln1[0], device), _f32(ln1[1], device))
        self.ln2 = (_f32(ln2[0], device), _f32(ln2[1], device))
        self.n_qkv, self.width = wqkv.shape
        self.n_fc = w_fc.shape[0]
        if fold_ln:
            self.wqkv, self.bqkv, self.cs_qkv = ops.fold_layernorm(wqkv.to(device), bqkv.to(device), self.ln1[0], self.ln1[1], dtype, device)
            self.w_fc, self.b_fc, self.cs_fc = ops.fold_layernorm(w_fc.to(device), b_fc.to(device), self.ln2[0], self.ln2[1], dtype, device)
        else:
            self.wqkv, self.bqkv = ops.pack_weight(wqkv.float(), dtype, device), _f32(bqkv, device)
            self.w_fc, self.b_fc = ops.pack_weight(w_fc.float(), dtype, device), _f32(b_fc, device)
        self.wo, self.bo = ops.pack_weight(wo.float(), dtype, device), _f32(bo, device)
        self.w_proj, self.b_proj = ops.pack_weight(w_proj.float(), dtype, device), _f32(b_proj, device)


def _clip_blocks(sd: Dict[str, Tensor], prefix: str, device, dtype, fold_ln: bool = False) -> List[_Block]:
    blocks = []
    i = 0
    while f"{prefix}resblocks.{i}.ln_1.weight" in sd:
        p = f"{prefix}resblocks.{i}."
        blocks.append(_Block(
            device, dtype, (sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]),
            sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
            sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"],
            (sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]),
            sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"], sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"], fold_ln=fold_ln))
        i += 1
    return blocks


class ClipDevice:
    """Frozen CLIP towers as OTTER uses them (``otter.py:177-222``)."""

    def __init__(self, sd: Dict[str, Tensor], arch: ClipArch, first_k: int, device, dtype=torch.float16,
                 prefix: str = "clip_model.", stream_dtype=None, text_dtype=None):
        """``dtype``: 16-bit MFMA operand type of the vision tower (> 99 % of the towers' work).  ``text_dtype``: operand type
        of the text tower, default fp16 whatever ``dtype`` is: rounding the text tower's tensors to bf16 costs 1.7e-3 of
        action error on its own (the 15 text tokens steer every softmax of the extraction), the whole vision tower in bf16
        3.6e-4 (measured class by class: scripts/bf16_error_attribution.py), and the text tower runs once per prompt."""
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"CLIP tower operand dtype must be float16 or bfloat16, got {dtype}")
        self.arch, self.first_k, self.device, self.dtype = arch, first_k, device, dtype
        self.text_dtype = text_dtype if text_dtype is not None else torch.float16
        if self.text_dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"text tower operand dtype must be float16 or bfloat16, got {self.text_dtype}")
        self.stream_dtype = stream_dtype if stream_dtype is not None else (torch.float16 if dtype == torch.float16 else torch.float32)
        if self.stream_dtype not in (torch.float32, dtype):
            raise ValueError(f"residual stream dtype must be float32 or {dtype}, got {self.stream_dtype}")
        v = prefix + "visual."
        self.act = ops.ACT_QUICKGELU if arch.activation == "quick_gelu" else ops.ACT_GELU
        conv = sd[v + "conv1.weight"].float()
        self.vw = conv.shape[0]
        self.patch = conv.shape[-1]
        self.k_im2col = ops.round_up(3 * self.patch * self.patch, ops.k_align(dtype))
        self.w_conv = ops.pack_weight(conv.reshape(self.vw, -1), dtype, device)
        self.cls = _f32(sd[v + "class_embedding"], device)
        self.vpos = _f32(sd[v + "positional_embedding"], device)
        self.ln_pre = (_f32(sd[v + "ln_pre.weight"], device), _f32(sd[v + "ln_pre.bias"], device))
        self.ln_post = (_f32(sd[v + "ln_post.weight"], device), _f32(sd[v + "ln_post.bias"], device))
        # vision tower: LayerNorm folded into the GEMMs when the residual stream has the operand type (the GEMM reads it)
        self.vblocks = _clip_blocks(sd, v + "transformer.", device, dtype, fold_ln=self.stream_dtype == dtype)
        proj = sd[v + "proj"].float()
        self.embed = proj.shape[1]
        self.w_vproj = ops.pack_weight(proj.t().contiguous(), dtype, device)
        self.v_heads = self.vw // 64
        # text tower
        self.tok_emb = _f32(sd[prefix + "token_embedding.weight"], device)
        self.tpos = _f32(sd[prefix + "positional_embedding"], device)
        self.tw = self.tok_emb.shape[1]
        self.t_heads = self.tw // 64
        self.tblocks = _clip_blocks(sd, prefix + "transformer.", device, self.text_dtype)
        # the text tower's residual stream: the vision tower's when the operand types agree, else fp32
        self.text_stream_dtype = self.stream_dtype if self.text_dtype == dtype else torch.float32
        self.ln_final = (_f32(sd[prefix + "ln_final.weight"], device), _f32(sd[prefix + "ln_final.bias"], device))
        self.w_tproj = ops.pack_weight(sd[prefix + "text_projection"].float().t().contiguous(), self.text_dtype, device)
        self._text_cache: Dict[bytes, Tensor] = {}
        # debugging aid (OTTER_DEBUG_STREAM=1): largest |value| of the residual stream after every block, so that a model whose
        # activations approach fp16's range (conversions saturate at +-65504 silently) can be spotted; costs a sync per block
        self.debug_stream = os.environ.get("OTTER_DEBUG_STREAM", "0") == "1"
        self.stream_absmax: List[float] = []

    # -- shared block body -------------------------------------------------------------------
    def _run_blocks(self, x: Tensor, blocks: List[_Block], n_seq: int, seq_len: int, heads: int, causal: bool,
                    stop_after_last_attention: bool) -> Tensor:
        last = None
        # With the LayerNorms folded into in_proj / c_fc, the (mean, rstd) of the stream's rows that those GEMMs need come out
        # of the residual product in front of them (one launch with its remainder rows' second pass); only the first block's
        # ln_1 reads a stream nobody has just written.
        stats = ops.row_stats(x) if (blocks and blocks[0].fold_ln) else None
        for i, b in enumerate(blocks):
            final = i == len(blocks) - 1
            if b.fold_ln:
                qkv = ops.gemm(x, b.wqkv, b.n_qkv, b.bqkv, ln_stats=stats, ln_colsum=b.cs_qkv)
            else:
                h = ops.layernorm(x, *b.ln1, out_dtype=b.dtype)
                qkv = ops.gemm(h, b.wqkv, b.n_qkv, b.bqkv)
            a = ops.attn_d64(qkv, n_seq, seq_len, heads, causal)
            if stop_after_last_attention and final:
                last = ops.gemm(a, b.wo, b.width, b.bo, out_f32=True)
                break
            if b.fold_ln and _FUSE_STATS:
                stats = ops.gemm_residual_stats_(a, b.wo, b.width, b.bo, x)
                m = ops.gemm(x, b.w_fc, b.n_fc, b.b_fc, act=self.act, ln_stats=stats, ln_colsum=b.cs_fc)
            elif b.fold_ln:
                ops.gemm(a, b.wo, b.width, b.bo, residual=x, out=x)
                m = ops.gemm(x, b.w_fc, b.n_fc, b.b_fc, act=self.act, ln_stats=ops.row_stats(x), ln_colsum=b.cs_fc)
            else:
                ops.gemm(a, b.wo, b.width, b.bo, residual=x, out=x)
                h = ops.layernorm(x, *b.ln2, out_dtype=b.dtype)
                m = ops.gemm(h, b.w_fc, b.n_fc, b.b_fc, act=self.act)
            if b.fold_ln and _FUSE_STATS and not final and blocks[i + 1].fold_ln:
                stats = ops.gemm_residual_stats_(m, b.w_proj, b.width, b.b_proj, x)
            else:
                ops.gemm(m, b.w_proj, b.width, b.b_proj, residual=x, out=x)
                stats = ops.row_stats(x) if (not final and blocks[i + 1].fold_ln) else None
            if self.debug_stream:
                self.stream_absmax.append(float(x.abs().max()))
        return last if stop_after_last_attention else x

    # -- vision ------------------------------------------------------------------------------
    def vision_tokens(self, images: Tensor) -> Tuple[Tensor, int, int]:
        """images (N,3,R,R) uint8 (normalisation fused) or fp32 (already normalised).

        Returns (tokens (N, 1+P, embed) in the tower dtype, un-normalised -- row 0 of each image is the class token --,
        P, embed).
        """
        n, _, res, _ = images.shape
        g = res // self.patch
        p = g * g
        if self.vpos.shape[0] != p + 1:
            raise ValueError(f"image size {res} gives {p} patches, positional embedding has {self.vpos.shape[0] - 1}")
        cols = ops.im2col_patches(images, self.patch, self.k_im2col, CLIP_MEAN, CLIP_STD, self.dtype)
        tok = ops.gemm(cols, self.w_conv, self.vw, None, out_f32=True)
        x = ops.vit_embed_lnpre(tok, self.cls, self.vpos, self.ln_pre[0], self.ln_pre[1], n, p, out_dtype=self.stream_dtype)
        last = self._run_blocks(x, self.vblocks, n, p + 1, self.v_heads, False, True)
        h = ops.layernorm(last, *self.ln_post, out_dtype=self.dtype)
        feats = ops.gemm(h, self.w_vproj, self.embed, None)
        return feats.view(n, p + 1, self.embed), p, self.embed

    # -- text --------------------------------------------------------------------------------
    def text_tokens(self, ids: Tensor, pooled_row: bool = False) -> Tensor:
        """ids (B, ctx) int32 on the device -> fp32 (B, first_k, embed), projected, un-normalised.

        With ``pooled_row`` one more row is appended per prompt: CLIP's pooled text feature, i.e. ``ln_final`` +
        ``text_projection`` of the end-of-text position (``encode_text``; the OpenCLIP extraction script adds it as a 16th
        query, ``script/open_clip_visualization.py:62-64,108-109``).  The tower then runs up to the last end-of-text
        position of the batch instead of ``first_k`` (exact under the causal mask either way).
        """
        b = ids.shape[0]
        k = self.first_k
        take = k
        eot = None
        if pooled_row:
            eot = ids.argmax(dim=-1)                                   # end-of-text has the highest id (49407)
            take = max(k, int(eot.max().item()) + 1)
        x = ops.text_embed(ids, take, self.tok_emb, self.tpos, out_dtype=self.text_stream_dtype)
        x = self._run_blocks(x, self.tblocks, b, take, self.t_heads, True, False)
        h = ops.layernorm(x, *self.ln_final, out_dtype=self.text_dtype)
        feats = ops.gemm(h, self.w_tproj, self.embed, None, out_f32=True).view(b, take, self.embed)
        if not pooled_row:
            return feats
        rows = torch.cat([torch.arange(k, device=ids.device).expand(b, k), eot[:, None]], dim=1)        # (B, k+1) positions
        return torch.gather(feats, 1, rows[:, :, None].expand(b, k + 1, self.embed)).contiguous()

    def text_tokens_cached(self, ids_host: Tensor) -> Tensor:
        """Host token rows (B, ctx) int32 -> device features, running the tower once per distinct prompt."""
        keys = [row.numpy().tobytes() for row in ids_host]
        missing = [i for i, kx in enumerate(keys) if kx not in self._text_cache]
        if missing:
            uniq: Dict[bytes, int] = {}
            for i in missing:
                uniq.setdefault(keys[i], i)
            rows = list(uniq.values())
            feats = self.text_tokens(ids_host[rows].to(self.device, non_blocking=True).contiguous())
            for j, i in enumerate(rows):
                self._text_cache[keys[i]] = feats[j]
            if len(self._text_cache) > 4096:
                for kx in list(self._text_cache)[: len(self._text_cache) - 4096]:
                    del self._text_cache[kx]
        return torch.stack([self._text_cache[kx] for kx in keys])


class PolicyDevice:
    """Causal policy transformer + action head (``otter/policy/transformer.py:53-160``, ``models.py:53-67``)."""

    def __init__(self, sd: Dict[str, Tensor], heads: int, device):
        self.device, self.heads = device, heads
        self.layers: List[_Block] = []
        i = 0
        while f"policy.layers.{i}.pre_ln.weight" in sd:
            p = f"policy.layers.{i}."
            wqkv = torch.cat([sd[p + f"attention.{n}.weight"].float() for n in ("query", "key", "value")], 0)
            bqkv = torch.cat([sd[p + f"attention.{n}.bias"].float() for n in ("query", "key", "value")], 0)
            self.layers.append(_Block(
                device, torch.float32, (sd[p + "pre_ln.weight"], sd[p + "pre_ln.bias"]), wqkv, bqkv,
                sd[p + "attention.out.weight"], sd[p + "attention.out.bias"],
                (sd[p + "post_ln.weight"], sd[p + "post_ln.bias"]),
                sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"], sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"]))
            i += 1
        self.dim = self.layers[0].width
        self.pe = _f32(sd["pe.pe"][0], device)                       # (seq_length, dim)
        self.w_h0, self.b_h0 = ops.pack_weight(sd["action_head.mlp.0.weight"].float(), torch.float32, device), _f32(sd["action_head.mlp.0.bias"], device)
        self.n_h0 = sd["action_head.mlp.0.weight"].shape[0]
        self.w_h2, self.b_h2 = ops.pack_weight(sd["action_head.mlp.2.weight"].float(), torch.float32, device), _f32(sd["action_head.mlp.2.bias"], device)
        self.n_out = sd["action_head.mlp.2.weight"].shape[0]

    def new_cache(self, batch: int, window: int) -> List[Tensor]:
        """Per-layer fp32 q|k|v rows (B, window, 3*dim): the GPU-resident K/V history."""
        return [torch.zeros(batch, window, 3 * self.dim, dtype=torch.float32, device=self.device) for _ in self.layers]

    def run(self, window: Tensor, cache: List[Tensor], n_query: int, pos0: int) -> Tensor:
        """Run positions pos0 .. pos0+n_query-1 of every episode; returns the fp32 outputs (B*n_query, dim).

        ``window`` (B, S, dim) holds the policy inputs; K/V of the processed positions are written into ``cache``
        and earlier positions are read from it.
        """
        b, s, _ = window.shape
        x = ops.add_pe(window, self.pe, n_query, pos0)
        f32 = torch.float32
        for blk, kv in zip(self.layers, cache):
            h = ops.layernorm(x, *blk.ln1, out_dtype=f32)
            if n_query == s:
                out = kv.view(b * s, 3 * self.dim)
            elif n_query == 1:
                out = kv[:, pos0, :]                                   # rows strided by S*3*dim
            else:
                raise ValueError("policy.run: n_query must be 1 (cached step) or the whole window")
            ops.gemm(h, blk.wqkv, blk.n_qkv, blk.bqkv, out=out)
            a = ops.attn_policy(kv, n_query, pos0, self.heads)
            ops.gemm(a, blk.wo, blk.width, blk.bo, residual=x, out=x)
            h = ops.layernorm(x, *blk.ln2, out_dtype=f32)
            m = ops.gemm(h, blk.w_fc, blk.n_fc, blk.b_fc, act=ops.ACT_GELU)
            ops.gemm(m, blk.w_proj, blk.width, blk.b_proj, residual=x, out=x)
        return x

    def run_rows(self, window: Tensor, cache: List[Tensor], idx: Tensor, lens: Tensor) -> Tensor:
        """The newest position of the episodes listed in ``idx`` (int32, device), each at its own window position
        ``lens[e] - 1`` (``lens``: int32 (B,), lengths after this step's append): only those rows are computed, their K/V are
        stored into ``cache`` and earlier positions are read from it.  Returns fp32 (len(idx), dim)."""
        x = ops.add_pe_rows(window, self.pe, idx, lens)
        f32 = torch.float32
        for blk, kv in zip(self.layers, cache):
            h = ops.layernorm(x, *blk.ln1, out_dtype=f32)
            fresh = ops.gemm(h, blk.wqkv, blk.n_qkv, blk.bqkv)
            a = ops.attn_policy_rows(fresh, kv, idx, lens, self.heads)
            ops.gemm(a, blk.wo, blk.width, blk.bo, residual=x, out=x)
            h = ops.layernorm(x, *blk.ln2, out_dtype=f32)
            m = ops.gemm(h, blk.w_fc, blk.n_fc, blk.b_fc, act=ops.ACT_GELU)
            ops.gemm(m, blk.w_proj, blk.width, blk.b_proj, residual=x, out=x)
        return x

    def head(self, x: Tensor, pick: Optional[Tensor] = None, group: int = 1, pick_bias: int = 0) -> Tensor:
        """x fp32 (R, dim) -> fp32 (R or R/group, horizon*action_dim); ``pick`` (+bias) selects one row per group."""
        a = ops.convert_rows(x, pick=pick, group=group, pick_bias=pick_bias) if pick is not None else x
        h = ops.gemm(a, self.w_h0, self.n_h0, self.b_h0, act=ops.ACT_RELU)
        return ops.gemm(h, self.w_h2, self.n_out, self.b_h2)


class OtterDevice:
    """All device state of one OTTER model: packed weights of the towers, the fusion modules and the policy."""

    def __init__(self, sd: Dict[str, Tensor], arch: ClipArch, *, first_k: int, pooling_heads: int, transformer_heads: int,
                 num_cameras: int, seq_length: int, action_horizon: int, action_dim: int, pool_true_text: bool, device,
                 clip_dtype=torch.float16, clip: Optional["ClipDevice"] = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ops.OtterHipError(f"OtterDevice needs a HIP device, got {device!r}; there is no CPU path")
        self.cams, self.first_k = num_cameras, first_k
        self.seq_length, self.horizon, self.action_dim = seq_length, action_horizon, action_dim
        self.pool_true_text = pool_true_text
        self.clip = clip if clip is not None else ClipDevice(sd, arch, first_k, self.device, clip_dtype)
        sdd = {k: v.detach().to(self.device) for k, v in sd.items() if not k.startswith("clip_model.")}
        self.text_pool = PoolingWeights(sdd, "text_pooling.", pooling_heads)
        self.vis_pool = [PoolingWeights(sdd, f"vision_poolings.{c}.", pooling_heads) for c in range(num_cameras)]
        self.temperature = torch.stack([sdd[f"visual_extraction.{c}.temperature"].float() for c in range(num_cameras)]).contiguous()
        self.pos_t = torch.stack([ops.pos_emb_pack(_f32(sdd[f"visual_extraction.{c}.pos_emb"], self.device), clip_dtype)
                                  for c in range(num_cameras)]).contiguous()
        self.prop = [(_f32(sdd[f"proprio_encoder.mlp.{i}.weight"].t(), self.device), _f32(sdd[f"proprio_encoder.mlp.{i}.bias"], self.device))
                     for i in (0, 2, 4)]
        self.d_text = self.text_pool.readouts * self.text_pool.d
        self.d_vis = self.vis_pool[0].readouts * self.vis_pool[0].d
        self.d_prop = self.prop[2][0].shape[1]
        self.f_t = self.d_text + self.cams * self.d_vis + self.d_prop
        w_in = sdd["input_projection.weight"].float()
        if w_in.shape[1] != self.f_t:
            raise ValueError(f"input_projection expects {w_in.shape[1]} features, modules produce {self.f_t}")
        self.w_in, self.b_in = ops.pack_weight(w_in, torch.float32, self.device), _f32(sdd["input_projection.bias"], self.device)
        self.k_in = self.w_in.shape[1]
        self.dim = w_in.shape[0]
        self.policy = PolicyDevice(sdd, transformer_heads, self.device)

    # ------------------------------------------------------------------ encoder (otter.py:224-267)
    def encode(self, images: Tensor, text_feats: Tensor, proprio: Tensor, t_rep: int, text_mask: Optional[Tensor] = None) -> Tensor:
        """images (M*cams, 3, R, R) uint8/fp32 ordered (obs, camera); text_feats fp32 (M/t_rep, K, D) un-normalised;
        proprio fp32 (M, P_in); text_mask bool (M/t_rep, K) or None.  Returns the policy input fp32 (M, dim)."""
        n = images.shape[0]
        m = n // self.cams
        b = m // t_rep
        k = self.first_k
        tokens, p, d = self.clip.vision_tokens(images)
        ta = ops.text_aware_pool(tokens, text_feats, self.pos_t, self.temperature, m, self.cams, t_rep, n_patches=p)
        if self.k_in == self.f_t:
            combined = torch.empty(m, self.f_t, dtype=torch.float32, device=self.device)
        else:                                    # feature count not a multiple of the GEMM's K-step: zero pad columns
            combined = torch.zeros(m, self.k_in, dtype=torch.float32, device=self.device)
        mask_b = mask_m = None
        if self.pool_true_text:
            if text_mask is None:
                raise ValueError("pool_true_text models need a text mask")
            mask_b = text_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mask_m = mask_b.repeat_interleave(t_rep, dim=0).contiguous() if t_rep > 1 else mask_b
        # the poolings of all cameras and of the text (otter.py:247,254): projections first, then ONE readout launch
        off = self.d_text
        problems = []
        for c in range(self.cams):
            problems.append(self.vis_pool[c].problem(ta[c].view(m * k, d), mask_m, combined[:, off:off + self.d_vis], m, k))
            off += self.d_vis
        tfn = ops.l2norm_rows(text_feats.view(b * k, d))
        tt = combined[:, : self.d_text] if t_rep == 1 else torch.empty(b, self.d_text, dtype=torch.float32, device=self.device)
        problems.append(self.text_pool.problem(tfn, mask_b, tt, b, k))
        for i in range(0, len(problems), 4):
            ops.readout_pool_multi(problems[i:i + 4])
        if t_rep != 1:
            ops.repeat_rows(tt, combined[:, : self.d_text], t_rep)
        h = ops.linear_f32(proprio, self.prop[0][0], self.prop[0][1], ops.ACT_RELU)
        h = ops.linear_f32(h, self.prop[1][0], self.prop[1][1], ops.ACT_RELU)
        ops.linear_f32(h, self.prop[2][0], self.prop[2][1], ops.ACT_NONE, out=combined[:, off:off + self.d_prop])
        return ops.gemm(combined, self.w_in, self.dim, self.b_in)

    # ------------------------------------------------------------------ full-sequence actions (otter.py:269-291)
    def actions(self, images: Tensor, text_feats: Tensor, proprio: Tensor, batch: int, steps: int,
                text_mask: Optional[Tensor] = None) -> Tensor:
        """Batched forward over (batch, steps) observations -> fp32 (batch, steps, horizon, action_dim)."""
        if steps > self.policy.pe.shape[0]:
            raise ValueError(f"sequence of {steps} steps exceeds the positional table ({self.policy.pe.shape[0]})")
        x = self.encode(images, text_feats, proprio, steps, text_mask).view(batch, steps, self.dim)
        cache = self.policy.new_cache(batch, steps)
        y = self.policy.run(x, cache, steps, 0)
        return self.policy.head(y).view(batch, steps, self.horizon, self.action_dim)
