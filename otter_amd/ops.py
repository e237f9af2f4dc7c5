"""Tensor-level wrappers over the C ABI.

torch is used for device memory and the current HIP stream only; every function here checks shapes,
dtypes, device and contiguity on the host (a kernel that reads out of bounds can take the GPU down) and then
passes raw pointers to ``libotter_hip.so``.  No function here computes anything in torch.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_RELU, DT_BF16, DT_F16, DT_F32, OUT_F32, RESIDUAL, TILE_128,  # noqa: F401
                   OtterHipError, ReadoutBlock)

Tensor = torch.Tensor
GEMM_N_ALIGN = 256          # weight rows are padded so that both the 128x128 and the 256x256 tile kernels can read whole tiles
GEMM_K_BYTES = 128          # a K-step of the GEMM is 128 bytes per row: 64 halves or 32 floats

DT = {torch.bfloat16: DT_BF16, torch.float16: DT_F16, torch.float32: DT_F32}


def k_align(dtype) -> int:
    return GEMM_K_BYTES // torch.empty((), dtype=dtype).element_size()


def _dt(dtype, name: str, allow_f32: bool = True) -> int:
    if dtype not in DT or (dtype == torch.float32 and not allow_f32):
        raise TypeError(f"{name}: unsupported dtype {dtype}")
    return DT[dtype]


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: Tensor, dtype, name: str, dims: Optional[int] = None) -> Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t).__name__}")
    if t.device.type != "cuda":
        raise OtterHipError(f"{name}: tensor is on {t.device}; the HIP path needs device memory (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if dims is not None and t.dim() != dims:
        raise ValueError(f"{name}: {t.dim()} dims, expected {dims}")
    if t.stride(-1) != 1 and t.numel() > 0 and t.shape[-1] != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")
    return t


def _rows_view(t: Tensor, name: str) -> tuple:
    """(rows, cols, ld) of a 2-D row-major view with arbitrary row stride."""
    if t.dim() != 2:
        raise ValueError(f"{name}: expected 2-D, got {tuple(t.shape)}")
    return t.shape[0], t.shape[1], (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def _ptr(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------------------- weight packing

def pack_weight(w: Tensor, dtype=torch.bfloat16, device=None) -> Tensor:
    """torch Linear weight (N, K) -> ``dtype`` (N rounded up to 128, K rounded up to the GEMM's K-step), zero padded."""
    n, k = w.shape
    out = torch.zeros(round_up(n, GEMM_N_ALIGN), round_up(k, k_align(dtype)), dtype=dtype, device=device or w.device)
    out[:n, :k] = w.to(device=out.device, dtype=dtype)
    return out


# ----------------------------------------------------------------------------------------- kernels

_GEMM_WS = {}
_SPLIT_K = os.environ.get("OTTER_SPLIT_K", "1") != "0"       # tuning knob: 0 = never take the split-K path


_GEMM_WS_GUARD = threading.Lock()
_WS_LOCAL = threading.local()


def new_gemm_workspace(device, n: int = 4096) -> Tensor:
    """A split-K scratch buffer large enough for products with up to ``n`` output columns."""
    return torch.empty(_lib.load().otter_gemm_workspace_floats(round_up(n, 64)), dtype=torch.float32, device=device)


@contextlib.contextmanager
def gemm_workspace(ws: Tensor):
    """Every :func:`gemm` this thread issues inside the block uses ``ws`` as its split-K scratch.  A captured HIP graph bakes
    the scratch address in, whatever stream it is later replayed on, so every graph owns a buffer of its own: two graphs
    replayed at the same time on different streams must not share partial products."""
    prev = getattr(_WS_LOCAL, "ws", None)
    _WS_LOCAL.ws = ws
    try:
        yield ws
    finally:
        _WS_LOCAL.ws = prev


def _gemm_workspace(device, n: int):
    """-> (scratch, lock or None).  Default: one buffer per (device, current stream) and a lock that makes the
    partial-product launch and the reduce launch of one call adjacent on that stream (ctypes releases the GIL during the
    call; without the lock two threads on one stream could interleave as A.partial, B.partial, A.reduce)."""
    need = _lib.load().otter_gemm_workspace_floats(round_up(n, 64))
    own = getattr(_WS_LOCAL, "ws", None)
    if own is not None and own.device == device:
        if own.numel() < need:
            raise ValueError(f"gemm: the scoped workspace holds {own.numel()} floats, N={n} needs {need}")
        return own, None
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    with _GEMM_WS_GUARD:
        entry = _GEMM_WS.get(key)
        if entry is None or entry[0].numel() < need:
            entry = _GEMM_WS[key] = (torch.empty(need, dtype=torch.float32, device=device), entry[1] if entry else threading.Lock())
    return entry


def gemm(a: Tensor, w_packed: Tensor, n: int, bias: Optional[Tensor] = None, *, act: int = ACT_NONE,
         out_f32: bool = False, residual: Optional[Tensor] = None, out: Optional[Tensor] = None,
         ln_stats: Optional[Tensor] = None, ln_colsum: Optional[Tensor] = None,
         force_tile_128: bool = False, split_k: bool = True) -> Tensor:
    """out[M, n] = epilogue(a[M, K] @ w[n, K]^T + bias); operands bf16 / fp16 / fp32 (``w_packed`` from
    :func:`pack_weight` with the same dtype); output the operand type, or fp32 (``out_f32``, an fp32 residual, fp32
    operands).  ``residual`` (fp32, or the operands' 16-bit type) is added after the bias and fixes the output type.
    ``split_k``: few-tile problems and the < 256 remainder rows of big 16-bit ones may be multiplied split-K through a
    scratch buffer (deterministic; rows can differ from the one-pass kernels in the last bit of the fp32 sum)."""
    dt = _dt(a.dtype, "gemm.a")
    _chk(a, a.dtype, "gemm.a")
    _chk(w_packed, a.dtype, "gemm.w", 2)
    m, k, lda = _rows_view(a, "gemm.a")
    if not w_packed.is_contiguous() or w_packed.shape[1] != k:
        raise ValueError(f"gemm: weight {tuple(w_packed.shape)} does not match K={k} (activations must carry the padded K)")
    if w_packed.shape[0] < round_up(n, GEMM_N_ALIGN):
        raise ValueError(f"gemm: packed weight has {w_packed.shape[0]} rows, needs {round_up(n, GEMM_N_ALIGN)}")
    if bias is not None:
        _chk(bias, torch.float32, "gemm.bias", 1)
        if bias.numel() < n:
            raise ValueError("gemm: bias shorter than N")
    if residual is not None and residual.dtype not in (torch.float32, a.dtype):
        raise ValueError(f"gemm.residual: {residual.dtype}, expected float32 or {a.dtype}")
    f32_out = out_f32 or (residual is not None and residual.dtype == torch.float32) or a.dtype == torch.float32
    epi = act | (OUT_F32 if f32_out else 0) | (RESIDUAL if residual is not None else 0) | (TILE_128 if force_tile_128 else 0)
    odt = torch.float32 if f32_out else a.dtype
    if out is None:
        out = torch.empty(m, n, dtype=odt, device=a.device)
    _chk(out, odt, "gemm.out")
    mo, no, ldc = _rows_view(out, "gemm.out")
    if mo != m or no != n:
        raise ValueError(f"gemm: out is {tuple(out.shape)}, expected ({m}, {n})")
    ldr = 0
    if residual is not None:
        _chk(residual, odt, "gemm.residual")
        mr, nr, ldr = _rows_view(residual, "gemm.residual")
        if mr != m or nr != n:
            raise ValueError("gemm: residual shape mismatch")
    if m == 0:
        return out
    if ln_stats is not None:
        # LayerNorm folded into the epilogue: w_packed holds W * gamma, bias holds bias + W . beta (see fold_layernorm)
        if residual is not None or ln_colsum is None:
            raise ValueError("gemm: ln_stats goes with ln_colsum and without a residual")
        _chk(ln_stats, torch.float32, "gemm.ln_stats", 2)
        _chk(ln_colsum, torch.float32, "gemm.ln_colsum", 1)
        if tuple(ln_stats.shape) != (m, 2) or ln_colsum.numel() < n or not ln_stats.is_contiguous():
            raise ValueError("gemm: ln_stats must be (M, 2) contiguous and ln_colsum at least N long")
    # scratch for the split-K pass over the remainder rows of a big 16-bit problem (one buffer per device and stream)
    ws, lock = (_gemm_workspace(a.device, n) if (split_k and _SPLIT_K and (m < 256 if a.dtype == torch.float32 else (m % 256 != 0 or m < 1024)))
                else (None, None))
    with lock if lock is not None else contextlib.nullcontext():
        _lib.call("otter_gemm_ws", dt, _ptr(a), lda, _ptr(w_packed), w_packed.shape[0], _ptr(bias), _ptr(residual), ldr,
                  _ptr(ln_stats), _ptr(ln_colsum), _ptr(out), ldc, m, n, k, epi, _ptr(ws), ws.numel() if ws is not None else 0, _stream())
    return out


def layernorm(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], *, eps: float = 1e-5,
              out_dtype=torch.bfloat16, out: Optional[Tensor] = None) -> Tensor:
    _chk(x, x.dtype, "layernorm.x")
    idt = _dt(x.dtype, "layernorm.x")
    rows, width, ldx = _rows_view(x, "layernorm.x")
    for t, nm in ((gamma, "gamma"), (beta, "beta")):
        if t is not None:
            _chk(t, torch.float32, "layernorm." + nm, 1)
            if t.numel() != width:
                raise ValueError(f"layernorm.{nm}: {t.numel()} elements, expected {width}")
    odt = out_dtype
    dt = _dt(odt, "layernorm.out_dtype")
    if out is None:
        out = torch.empty(rows, width, dtype=odt, device=x.device)
    _chk(out, odt, "layernorm.out")
    ro, wo, ldy = _rows_view(out, "layernorm.out")
    if ro != rows or wo != width:
        raise ValueError("layernorm: out shape mismatch")
    if rows == 0:
        return out
    _lib.call("otter_layernorm", _ptr(x), idt, ldx, _ptr(gamma), _ptr(beta), _ptr(out), ldy, rows, width, eps, dt, _stream())
    return out


def row_stats(x: Tensor, eps: float = 1e-5) -> Tensor:
    """(rows, 2) fp32: mean and 1/sqrt(var + eps) of every row of x (fp32 or 16-bit)."""
    _chk(x, x.dtype, "row_stats.x")
    rows, width, ldx = _rows_view(x, "row_stats.x")
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    if rows:
        _lib.call("otter_row_stats", _ptr(x), _dt(x.dtype, "row_stats.x"), ldx, _ptr(stats), rows, width, eps, _stream())
    return stats


def gemm_residual_stats_(a: Tensor, w_packed: Tensor, n: int, bias: Optional[Tensor], x: Tensor, eps: float = 1e-5) -> Tensor:
    """x[M, n] += a[M, K] @ w[n, K]^T + bias in place on the 16-bit residual stream ``x``; returns the (M, 2) fp32 row
    statistics (mean, 1/sqrt(var + eps)) of the updated rows.  The same results as ``gemm(a, w, n, bias, residual=x, out=x)``
    followed by ``row_stats(x)``, with one launch less when the remainder rows of a big problem go split-K."""
    dt = _dt(a.dtype, "gemm_residual_stats.a", False)
    _chk(a, a.dtype, "gemm_residual_stats.a")
    _chk(w_packed, a.dtype, "gemm_residual_stats.w", 2)
    _chk(x, a.dtype, "gemm_residual_stats.x")
    m, k, lda = _rows_view(a, "gemm_residual_stats.a")
    mx, nx, ldx = _rows_view(x, "gemm_residual_stats.x")
    if not w_packed.is_contiguous() or w_packed.shape[1] != k or w_packed.shape[0] < round_up(n, GEMM_N_ALIGN):
        raise ValueError(f"gemm_residual_stats: weight {tuple(w_packed.shape)} does not match K={k}, N={n}")
    if mx != m or nx != n:
        raise ValueError(f"gemm_residual_stats: x is {tuple(x.shape)}, expected ({m}, {n})")
    if bias is not None:
        _chk(bias, torch.float32, "gemm_residual_stats.bias", 1)
        if bias.numel() < n:
            raise ValueError("gemm_residual_stats: bias shorter than N")
    stats = torch.empty(m, 2, dtype=torch.float32, device=x.device)
    if m == 0:
        return stats
    ws, lock = _gemm_workspace(a.device, n) if (_SPLIT_K and (m % 256 != 0 or m < 1024)) else (None, None)
    with lock if lock is not None else contextlib.nullcontext():
        _lib.call("otter_gemm_residual_stats", dt, _ptr(a), lda, _ptr(w_packed), w_packed.shape[0], _ptr(bias), _ptr(x), ldx,
                  m, n, k, _ptr(stats), eps, _ptr(ws), ws.numel() if ws is not None else 0, _stream())
    return stats


def fold_layernorm(weight: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, dtype, device=None):
    """Fold ``LayerNorm(gamma, beta)`` into the Linear that follows it: (packed W * gamma in ``dtype``, bias + W . beta,
    column sums of the ROUNDED W * gamma) -- the operands of ``gemm(..., ln_stats=row_stats(x), ln_colsum=...)``."""
    w = weight.detach().float()
    wg = w * gamma.detach().float()[None, :]
    packed = pack_weight(wg, dtype, device)
    colsum = packed.float().sum(dim=1).contiguous()                      # over the padded K too (padding is zero)
    b = w @ beta.detach().float()
    if bias is not None:
        b = b + bias.detach().float()
    return packed, b.to(packed.device).contiguous(), colsum


def vit_embed_lnpre(patch_tok: Tensor, cls: Tensor, pos: Tensor, gamma: Tensor, beta: Tensor, n_img: int, n_patches: int,
                    eps: float = 1e-5, out_dtype=torch.float32) -> Tensor:
    _chk(patch_tok, torch.float32, "vit_embed.patch_tok", 2)
    rows, width, ld = _rows_view(patch_tok, "vit_embed.patch_tok")
    if rows != n_img * n_patches:
        raise ValueError(f"vit_embed: {rows} patch rows, expected {n_img * n_patches}")
    _chk(cls, torch.float32, "vit_embed.cls", 1)
    _chk(pos, torch.float32, "vit_embed.pos", 2)
    if cls.numel() != width or tuple(pos.shape) != (n_patches + 1, width) or not pos.is_contiguous():
        raise ValueError("vit_embed: class/positional embedding shape mismatch")
    for t in (gamma, beta):
        _chk(t, torch.float32, "vit_embed.ln", 1)
        if t.numel() != width:
            raise ValueError("vit_embed: ln_pre width mismatch")
    x = torch.empty(n_img * (n_patches + 1), width, dtype=out_dtype, device=patch_tok.device)
    _lib.call("otter_vit_embed_lnpre", _ptr(patch_tok), ld, _ptr(cls), _ptr(pos), _ptr(gamma), _ptr(beta), _ptr(x),
              _dt(out_dtype, "vit_embed.out_dtype"), width,
              n_img, n_patches, width, eps, _stream())
    return x


def text_embed(ids: Tensor, take: int, token_embedding: Tensor, pos: Tensor, out_dtype=torch.float32) -> Tensor:
    _chk(ids, torch.int32, "text_embed.ids", 2)
    if not ids.is_contiguous():
        raise ValueError("text_embed.ids must be contiguous")
    b, ctx = ids.shape
    _chk(token_embedding, torch.float32, "text_embed.table", 2)
    _chk(pos, torch.float32, "text_embed.pos", 2)
    vocab, width = token_embedding.shape
    if not token_embedding.is_contiguous() or not pos.is_contiguous() or pos.shape[1] != width or pos.shape[0] < take:
        raise ValueError("text_embed: table/positional embedding shape mismatch")
    if not (0 < take <= ctx):
        raise ValueError(f"text_embed: take={take} outside 1..{ctx}")
    x = torch.empty(b * take, width, dtype=out_dtype, device=ids.device)
    _lib.call("otter_text_embed", _ptr(ids), ctx, take, _ptr(token_embedding), vocab, _ptr(pos), _ptr(x), _dt(out_dtype, "text_embed.out_dtype"),
              width, b, width, _stream())
    return x


def im2col_patches(images: Tensor, patch: int, kpad: int, mean: Sequence[float], std: Sequence[float],
                   out_dtype=torch.bfloat16) -> Tensor:
    """images (N, 3, R, R) uint8 (normalisation fused) or float32 (already normalised) -> bf16/fp16 (N*g*g, kpad)."""
    if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != images.shape[3]:
        raise ValueError(f"im2col: expected (N, 3, R, R), got {tuple(images.shape)}")
    if images.dtype not in (torch.uint8, torch.float32):
        raise TypeError(f"im2col: dtype {images.dtype}; uint8 or float32")
    _chk(images, images.dtype, "im2col.images")
    if not images.is_contiguous():
        raise ValueError("im2col: images must be contiguous")
    n, _, res, _ = images.shape
    g = res // patch
    out = torch.empty(n * g * g, kpad, dtype=out_dtype, device=images.device)
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    _lib.call("otter_im2col_patches", _dt(out_dtype, "im2col.out_dtype", False), _ptr(images), int(images.dtype == torch.uint8), _ptr(out), n, res, patch, kpad, m3, s3, _stream())
    return out


def attn_d64(qkv: Tensor, n_seq: int, seq_len: int, heads: int, causal: bool) -> Tensor:
    dt = _dt(qkv.dtype, "attn.qkv", False)
    _chk(qkv, qkv.dtype, "attn.qkv", 2)
    rows, width, ld = _rows_view(qkv, "attn.qkv")
    if rows != n_seq * seq_len or width != 3 * heads * 64:
        raise ValueError(f"attn: qkv {tuple(qkv.shape)} does not match n_seq={n_seq} L={seq_len} heads={heads}")
    out = torch.empty(rows, heads * 64, dtype=qkv.dtype, device=qkv.device)
    if rows == 0:
        return out
    _lib.call("otter_attn_fwd_d64", dt, _ptr(qkv), ld, _ptr(out), heads * 64, n_seq, seq_len, heads, int(causal), _stream())
    return out


def attn_policy(qkv_cache: Tensor, n_query: int, pos0: int, heads: int) -> Tensor:
    """qkv_cache (B, S, 3*dim) fp32 -> fp32 (B*n_query, dim)."""
    _chk(qkv_cache, torch.float32, "attn_policy.qkv", 3)
    if not qkv_cache.is_contiguous():
        raise ValueError("attn_policy: cache must be contiguous")
    b, s, w3 = qkv_cache.shape
    dim = w3 // 3
    if dim * 3 != w3 or dim % heads != 0:
        raise ValueError("attn_policy: bad cache width")
    if pos0 < 0 or pos0 + n_query > s:
        raise ValueError(f"attn_policy: positions {pos0}..{pos0 + n_query - 1} outside window {s}")
    out = torch.empty(b * n_query, dim, dtype=torch.float32, device=qkv_cache.device)
    _lib.call("otter_attn_policy", _ptr(qkv_cache), w3, _ptr(out), dim, b, s, n_query, pos0, heads, dim // heads, _stream())
    return out


def pos_emb_pack(pos: Tensor, dtype=torch.bfloat16) -> Tensor:
    """pos_emb (P, D) fp32 -> (ceil(P/16)*16*D,) 16-bit table in the order the pooling kernel's lanes read it."""
    _chk(pos, torch.float32, "pos_emb", 2)
    if not pos.is_contiguous():
        raise ValueError("pos_emb must be contiguous")
    p, d = pos.shape
    out = torch.empty(-(-p // 16) * 16 * d, dtype=dtype, device=pos.device)
    _lib.call("otter_pos_emb_pack", _dt(dtype, "pos_emb_pack", False), _ptr(pos), _ptr(out), p, d, _stream())
    return out


def text_aware_pool(patches: Tensor, text: Tensor, pos_packed: Tensor, temperature: Tensor, n_obs: int, cams: int, t_rep: int,
                    n_patches: Optional[int] = None, out_dtype=torch.float32) -> Tensor:
    """Fused text-aware visual extraction.

    patches bf16/fp16: either (n_obs*cams, P, D) contiguous, or (n_obs*cams, L, D) with ``n_patches`` = P < L given,
    in which case the LAST P rows of each unit are used (skips the class token of a ViT output);
    text (n_obs/t_rep, K, D) fp32 or the patches' dtype; pos_packed (cams, ceil(P/16)*16*D) from :func:`pos_emb_pack` in
    the patches' dtype; -> (cams, n_obs, K, D) in ``out_dtype`` (fp32 or the patches' dtype).
    """
    dt = _dt(patches.dtype, "tave.patches", False)
    _chk(patches, patches.dtype, "tave.patches", 3)
    if text.dtype not in (torch.float32, patches.dtype) or out_dtype not in (torch.float32, patches.dtype):
        raise TypeError(f"tave: text {text.dtype} / out {out_dtype} must be float32 or {patches.dtype}")
    _chk(text, text.dtype, "tave.text", 3)
    _chk(pos_packed, patches.dtype, "tave.pos_packed", 2)
    _chk(temperature, torch.float32, "tave.temperature", 1)
    units, rows, d = patches.shape
    p = n_patches if n_patches is not None else rows
    nt, k, d2 = text.shape
    if units != n_obs * cams or d2 != d or n_obs % t_rep != 0 or nt != n_obs // t_rep or p > rows:
        raise ValueError(f"tave: patches {tuple(patches.shape)}, text {tuple(text.shape)}, n_obs={n_obs}, cams={cams}, t_rep={t_rep}")
    if tuple(pos_packed.shape) != (cams, -(-p // 16) * 16 * d) or temperature.numel() != cams:
        raise ValueError("tave: pos_packed / temperature shape mismatch")
    for t in (patches, text, pos_packed, temperature):
        if not t.is_contiguous():
            raise ValueError("tave: operands must be contiguous")
    out = torch.empty(cams, n_obs, k, d, dtype=out_dtype, device=patches.device)
    if n_obs == 0:
        return out
    base = ctypes.c_void_p(patches.data_ptr() + (rows - p) * d * 2)
    io16 = int(text.dtype != torch.float32) | (int(out_dtype != torch.float32) << 1)
    if io16 == 0:
        _lib.call("otter_text_aware_pool_fwd", dt, base, rows * d, _ptr(text), _ptr(pos_packed), _ptr(temperature), _ptr(out),
                  n_obs, cams, p, d, k, t_rep, _stream())
    else:
        _lib.call("otter_text_aware_pool_fwd_io", dt, base, rows * d, _ptr(text), _ptr(pos_packed), _ptr(temperature), _ptr(out),
                  n_obs, cams, p, d, k, t_rep, io16, _stream())
    return out


def _readout_check(query: Tensor, final_w: Tensor, final_b: Tensor, kv: Tensor, mask: Optional[Tensor], out: Tensor,
                   n_obs: int, nk: int, readouts: int, d: int):
    """Shape / type checks of one pooling; -> (ldkv, ldo)."""
    _chk(kv, torch.float32, "readout.kv", 2)
    rows, _, ldkv = _rows_view(kv, "readout.kv")
    if rows != n_obs * nk:
        raise ValueError("readout: kv rows mismatch")
    _chk(out, torch.float32, "readout.out", 2)
    ro, wo, ldo = _rows_view(out, "readout.out")
    if ro != n_obs or wo != readouts * d:
        raise ValueError("readout: out shape mismatch")
    if mask is not None:
        _chk(mask, torch.uint8, "readout.mask", 2)
        if tuple(mask.shape) != (n_obs, nk) or not mask.is_contiguous():
            raise ValueError("readout: mask shape mismatch")
    for t, nme, cnt in ((query, "query", readouts * d), (final_w, "final_w", d), (final_b, "final_b", d)):
        _chk(t, torch.float32, "readout." + nme)
        if t.numel() != cnt or not t.is_contiguous():
            raise ValueError(f"readout.{nme}: bad size")
    return ldkv, ldo


READOUT_FIELDS = ("q_ln_w", "q_ln_b", "wq_t", "bq", "wo_t", "bo", "ln_w", "ln_b", "fc1_t", "fc1_b", "fc2_t", "fc2_b")   # otter_readout_block
_BLOCK_ARRAYS: dict = {}


def readout_blocks(params):
    """ctypes array of ``otter_readout_block`` over 12 fp32 device tensors per block (order: ``READOUT_FIELDS``).  Built once
    per set of addresses and kept together with the tensors, so that the raw pointers in the structs stay valid.
    A ctypes array passes through unchanged."""
    if not isinstance(params, (list, tuple)):
        return params
    key = tuple(t.data_ptr() for t in params)
    hit = _BLOCK_ARRAYS.get(key)
    if hit is None:
        nf = len(READOUT_FIELDS)
        if len(params) == 0 or len(params) % nf != 0:
            raise ValueError(f"readout: {len(params)} block tensors, expected a positive multiple of {nf}")
        arr = (ReadoutBlock * (len(params) // nf))()
        for i, t in enumerate(params):
            _chk(t, torch.float32, "readout.block." + READOUT_FIELDS[i % nf])
            if not t.is_contiguous():
                raise ValueError("readout: block parameters must be contiguous")
            setattr(arr[i // nf], READOUT_FIELDS[i % nf], ctypes.c_void_p(t.data_ptr()))
        if len(_BLOCK_ARRAYS) > 1024:
            _BLOCK_ARRAYS.clear()
        hit = _BLOCK_ARRAYS[key] = (arr, list(params))
    return hit[0]


def readout_pool(blocks, query: Tensor, final_w: Tensor, final_b: Tensor, kv: Tensor, mask: Optional[Tensor], out: Tensor,
                 n_obs: int, nk: int, readouts: int, d: int, heads: int, hidden: int) -> Tensor:
    """``blocks``: the blocks' tensors (12 each, ``READOUT_FIELDS`` order) or a ctypes array of :class:`ReadoutBlock`;
    kv fp32 (n_obs*nk, >= n_blocks*2d); out fp32 view (n_obs, readouts*d)."""
    blocks = readout_blocks(blocks)
    ldkv, ldo = _readout_check(query, final_w, final_b, kv, mask, out, n_obs, nk, readouts, d)
    _lib.call("otter_readout_pool_fwd", blocks, len(blocks), _ptr(query), _ptr(final_w), _ptr(final_b), _ptr(kv), ldkv, _ptr(mask),
              _ptr(out), ldo, n_obs, nk, readouts, d, heads, hidden, _stream())
    return out


def readout_pool_multi(problems) -> None:
    """Several poolings in one launch.  ``problems``: sequence of tuples with the arguments of :func:`readout_pool`, in
    its order (blocks, query, final_w, final_b, kv, mask, out, n_obs, nk, readouts, d, heads, hidden); at most 4."""
    arr = (_lib.ReadoutProblem * len(problems))()
    for q, (blocks, query, final_w, final_b, kv, mask, out, n_obs, nk, readouts, d, heads, hidden) in zip(arr, problems):
        blocks = readout_blocks(blocks)
        ldkv, ldo = _readout_check(query, final_w, final_b, kv, mask, out, n_obs, nk, readouts, d)
        q.blocks, q.n_blocks = ctypes.cast(blocks, ctypes.POINTER(_lib.ReadoutBlock)), len(blocks)
        q.query, q.final_ln_w, q.final_ln_b = query.data_ptr(), final_w.data_ptr(), final_b.data_ptr()
        q.kv, q.ldkv, q.mask = kv.data_ptr(), ldkv, (None if mask is None else mask.data_ptr())
        q.out, q.ldo = out.data_ptr(), ldo
        q.n_obs, q.nk, q.readouts, q.d, q.heads, q.hidden = n_obs, nk, readouts, d, heads, hidden
    _lib.call("otter_readout_pool_fwd_multi", arr, len(problems), _stream())


def l2norm_rows(x: Tensor) -> Tensor:
    """fp32 / bf16 / fp16 (rows, width) -> fp32 rows of unit L2 norm."""
    dt = _dt(x.dtype, "l2norm.x")
    _chk(x, x.dtype, "l2norm.x", 2)
    rows, width, ldx = _rows_view(x, "l2norm.x")
    y = torch.empty(rows, width, dtype=torch.float32, device=x.device)
    _lib.call("otter_l2norm_rows", dt, _ptr(x), ldx, _ptr(y), width, rows, width, _stream())
    return y


def linear_f32(x: Tensor, wt: Tensor, bias: Optional[Tensor], act: int = ACT_NONE, out: Optional[Tensor] = None) -> Tensor:
    """fp32 small linear: x (M, K), wt (K, N) transposed weight."""
    _chk(x, torch.float32, "linear_f32.x", 2)
    _chk(wt, torch.float32, "linear_f32.wt", 2)
    m, k, ldx = _rows_view(x, "linear_f32.x")
    if wt.shape[0] != k or not wt.is_contiguous():
        raise ValueError("linear_f32: weight shape mismatch")
    n = wt.shape[1]
    if bias is not None:
        _chk(bias, torch.float32, "linear_f32.bias", 1)
        if bias.numel() != n:
            raise ValueError("linear_f32: bias size mismatch")
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=x.device)
    _chk(out, torch.float32, "linear_f32.out", 2)
    mo, no, ldy = _rows_view(out, "linear_f32.out")
    if mo != m or no != n:
        raise ValueError("linear_f32: out shape mismatch")
    _lib.call("otter_linear_f32", _ptr(x), ldx, _ptr(wt), _ptr(bias), _ptr(out), ldy, m, n, k, act, _stream())
    return out


def convert_rows(src: Tensor, out_dtype=torch.float32, cols_pad: Optional[int] = None, rep: int = 1,
                 pick: Optional[Tensor] = None, group: int = 1, rows: Optional[int] = None, pick_bias: int = 0) -> Tensor:
    """fp32 (R, C) -> ``out_dtype`` (rows, cols_pad); see ``otter_convert_rows`` for the row maps."""
    _chk(src, torch.float32, "cast.src", 2)
    r, c, lds = _rows_view(src, "cast.src")
    cols_pad = cols_pad or c
    if pick is not None:
        _chk(pick, torch.int32, "cast.pick", 1)
        n_rows = pick.numel()
        if n_rows * group != r:
            raise ValueError("cast: pick/group do not match the source rows")
    else:
        n_rows = rows if rows is not None else r * rep
        if (n_rows - 1) // rep >= r:
            raise ValueError("cast: rep maps past the source rows")
    dst = torch.empty(n_rows, cols_pad, dtype=out_dtype, device=src.device)
    _lib.call("otter_convert_rows", _dt(out_dtype, "convert.out_dtype"), _ptr(src), lds, _ptr(dst), cols_pad, n_rows, c, cols_pad, rep, _ptr(pick), group, pick_bias, _stream())
    return dst


def repeat_rows(src: Tensor, dst: Tensor, rep: int) -> Tensor:
    _chk(src, torch.float32, "repeat.src", 2)
    _chk(dst, torch.float32, "repeat.dst", 2)
    r, c, lds = _rows_view(src, "repeat.src")
    rd, cd, ldd = _rows_view(dst, "repeat.dst")
    if cd != c or rd != r * rep:
        raise ValueError("repeat_rows: shape mismatch")
    _lib.call("otter_repeat_rows", _ptr(src), lds, _ptr(dst), ldd, rd, c, rep, _stream())
    return dst


def add_pe(window: Tensor, pe: Tensor, n_query: int, pos0: int) -> Tensor:
    _chk(window, torch.float32, "add_pe.window", 3)
    _chk(pe, torch.float32, "add_pe.pe", 2)
    b, s, dim = window.shape
    if not window.is_contiguous() or not pe.is_contiguous() or pe.shape[1] != dim or pe.shape[0] < pos0 + n_query:
        raise ValueError("add_pe: shape mismatch")
    out = torch.empty(b * n_query, dim, dtype=torch.float32, device=window.device)
    _lib.call("otter_add_pe", _ptr(window), _ptr(pe), _ptr(out), b, s, n_query, pos0, dim, _stream())
    return out


def _rows_index(idx: Tensor, lens: Tensor, n_episodes: int, name: str) -> int:
    _chk(idx, torch.int32, name + ".idx", 1)
    _chk(lens, torch.int32, name + ".lens", 1)
    if lens.numel() != n_episodes or not idx.is_contiguous() or not lens.is_contiguous():
        raise ValueError(f"{name}: lens must hold one length per episode ({n_episodes}), idx / lens contiguous")
    return idx.numel()


def add_pe_rows(window: Tensor, pe: Tensor, idx: Tensor, lens: Tensor) -> Tensor:
    """Newest step of each listed episode plus its positional row: window (B, S, dim), idx int32 (n,) episode numbers,
    lens int32 (B,) lengths after this step's append -> fp32 (n, dim)."""
    _chk(window, torch.float32, "add_pe_rows.window", 3)
    _chk(pe, torch.float32, "add_pe_rows.pe", 2)
    b, s, dim = window.shape
    n = _rows_index(idx, lens, b, "add_pe_rows")
    if not window.is_contiguous() or not pe.is_contiguous() or pe.shape[1] != dim or pe.shape[0] < s:
        raise ValueError("add_pe_rows: shape mismatch (the positional table must cover the window)")
    out = torch.empty(n, dim, dtype=torch.float32, device=window.device)
    _lib.call("otter_add_pe_rows", _ptr(window), _ptr(pe), _ptr(out), _ptr(idx), _ptr(lens), n, b, s, dim, _stream())
    return out


def attn_policy_rows(fresh: Tensor, qkv_cache: Tensor, idx: Tensor, lens: Tensor, heads: int) -> Tensor:
    """fresh fp32 (n, 3*dim) q|k|v of the newest step of episodes idx; qkv_cache (B, S, 3*dim) receives them at each
    episode's position lens[e] - 1 -> attention outputs fp32 (n, dim)."""
    _chk(fresh, torch.float32, "attn_policy_rows.fresh", 2)
    _chk(qkv_cache, torch.float32, "attn_policy_rows.cache", 3)
    b, s, w3 = qkv_cache.shape
    dim = w3 // 3
    n = _rows_index(idx, lens, b, "attn_policy_rows")
    if not qkv_cache.is_contiguous() or dim * 3 != w3 or dim % heads != 0 or tuple(fresh.shape) != (n, w3):
        raise ValueError(f"attn_policy_rows: cache {tuple(qkv_cache.shape)}, fresh {tuple(fresh.shape)}, {n} rows, {heads} heads")
    _, _, ldf = _rows_view(fresh, "attn_policy_rows.fresh")
    out = torch.empty(n, dim, dtype=torch.float32, device=fresh.device)
    _lib.call("otter_attn_policy_rows", _ptr(fresh), ldf, _ptr(qkv_cache), w3, _ptr(out), dim, _ptr(idx), _ptr(lens), n, b, s, heads, dim // heads, _stream())
    return out


def window_append(src: Tensor, fresh: Tensor, dst: Tensor, lens_in: Tensor, lens_out: Tensor) -> None:
    _chk(src, torch.float32, "window.src", 3)
    _chk(dst, torch.float32, "window.dst", 3)
    _chk(fresh, torch.float32, "window.fresh", 2)
    _chk(lens_in, torch.int32, "window.lens_in", 1)
    _chk(lens_out, torch.int32, "window.lens_out", 1)
    b, s, dim = src.shape
    if dst.shape != src.shape or tuple(fresh.shape) != (b, dim) or lens_in.numel() != b or lens_out.numel() != b:
        raise ValueError("window_append: shape mismatch")
    for t in (src, dst, fresh):
        if not t.is_contiguous():
            raise ValueError("window_append: operands must be contiguous")
    _lib.call("otter_window_append", _ptr(src), _ptr(fresh), _ptr(dst), _ptr(lens_in), _ptr(lens_out), b, s, dim, _stream())


def action_step(delta_chunks: Tensor, cur_pose: Tensor, ring: Tensor, counts: Tensor, exec_horizon: int, receding: bool,
                k: float = 0.05, want_abs: bool = False):
    """One control step of many episodes: delta chunks fp32 (E, H, 10) + current poses fp64 (E, 10) ->
    (robot actions fp64 (E, 7), executed 10-d actions fp64 (E, 10), absolute chunks fp64 (E, H, 10) or None).
    ``ring`` fp64 (E, X, X, 10) and ``counts`` int32 (E,) are the queue state (see ``otter_action_step``)."""
    _chk(delta_chunks, torch.float32, "action_step.delta", 3)
    _chk(cur_pose, torch.float64, "action_step.pose", 2)
    _chk(ring, torch.float64, "action_step.ring", 4)
    _chk(counts, torch.int32, "action_step.counts", 1)
    e, h, a = delta_chunks.shape
    x = int(exec_horizon)
    if tuple(cur_pose.shape) != (e, a) or tuple(ring.shape) != (e, x, x, a) or counts.numel() != e:
        raise ValueError(f"action_step: delta {tuple(delta_chunks.shape)}, pose {tuple(cur_pose.shape)}, ring {tuple(ring.shape)}, counts {tuple(counts.shape)}")
    for t in (delta_chunks, cur_pose, ring, counts):
        if not t.is_contiguous():
            raise ValueError("action_step: operands must be contiguous")
    dev = delta_chunks.device
    robot = torch.empty(e, 7, dtype=torch.float64, device=dev)
    act10 = torch.empty(e, a, dtype=torch.float64, device=dev)
    abs_chunks = torch.empty(e, h, a, dtype=torch.float64, device=dev) if want_abs else None
    _lib.call("otter_action_step", _ptr(delta_chunks), _ptr(cur_pose), _ptr(abs_chunks), _ptr(ring), _ptr(counts), _ptr(act10), _ptr(robot),
              e, h, a, x, int(bool(receding)), float(k), _stream())
    return robot, act10, abs_chunks


# ------------------------------------------------------------------------------------------------------ training step
def _f32_2d(t: Tensor, name: str):
    _chk(t, torch.float32, name, 2)
    return _rows_view(t, name)


def sgemm(a: Tensor, b: Tensor, out: Tensor, *, trans_a: bool = False, trans_b: bool = False, bias: Optional[Tensor] = None,
          accumulate: bool = False, a_rowsum: Optional[Tensor] = None) -> Tensor:
    """out (+)= op(a) @ op(b) (+ bias), fp32, 2-D row-major views with arbitrary row strides (no packing).
    ``a_rowsum`` (fp32, M elements, contiguous): also (+)= the row sums of op(a) -- the bias gradient when op(a) = dy^T."""
    ra, ca, lda = _f32_2d(a, "sgemm.a")
    rb, cb, ldb = _f32_2d(b, "sgemm.b")
    m, k = (ca, ra) if trans_a else (ra, ca)
    kb, n = (cb, rb) if trans_b else (rb, cb)
    ro, co, ldc = _f32_2d(out, "sgemm.out")
    if k != kb or ro != m or co != n:
        raise ValueError(f"sgemm: op(a) is ({m},{k}), op(b) is ({kb},{n}), out is ({ro},{co})")
    if bias is not None:
        _chk(bias, torch.float32, "sgemm.bias", 1)
        if bias.numel() != n:
            raise ValueError("sgemm: bias length mismatch")
    sam, sak = (1, lda) if trans_a else (lda, 1)
    sbk, sbn = (1, ldb) if trans_b else (ldb, 1)
    if a_rowsum is not None:
        _chk(a_rowsum, torch.float32, "sgemm.a_rowsum", 1)
        if a_rowsum.numel() != m or not a_rowsum.is_contiguous():
            raise ValueError("sgemm: a_rowsum must be a contiguous vector of M elements")
    _lib.call("otter_sgemm_strided", _ptr(a), sam, sak, _ptr(b), sbk, sbn, _ptr(bias), _ptr(out), ldc, m, n, k, int(accumulate),
              _ptr(a_rowsum), _stream())
    return out


def act_fwd(z: Tensor, act: int) -> Tensor:
    _chk(z, torch.float32, "act.z")
    if not z.is_contiguous():
        raise ValueError("act_fwd: contiguous input expected")
    y = torch.empty_like(z)
    _lib.call("otter_act_fwd", _ptr(z), _ptr(y), z.numel(), act, _stream())
    return y


def act_bwd(z: Tensor, dy: Tensor, dz: Tensor, act: int) -> None:
    for t, nm in ((z, "z"), (dy, "dy"), (dz, "dz")):
        _chk(t, torch.float32, "act_bwd." + nm)
        if not t.is_contiguous() or t.numel() != z.numel():
            raise ValueError("act_bwd: contiguous operands of equal size expected")
    _lib.call("otter_act_bwd", _ptr(z), _ptr(dy), _ptr(dz), z.numel(), act, _stream())


def col_sum(x: Tensor, out: Tensor, y: Optional[Tensor] = None, accumulate: bool = True) -> Tensor:
    """out[c] (+)= sum_r x[r, c] * (y[r, c] if y is given else 1)."""
    r, c, ldx = _f32_2d(x, "col_sum.x")
    ldy = 0
    if y is not None:
        ry, cy, ldy = _f32_2d(y, "col_sum.y")
        if (ry, cy) != (r, c):
            raise ValueError("col_sum: y shape mismatch")
    _chk(out, torch.float32, "col_sum.out")
    if out.numel() != c or not out.is_contiguous():
        raise ValueError("col_sum: out must be contiguous with one element per column")
    _lib.call("otter_col_sum", _ptr(x), ldx, _ptr(y), ldy, r, c, _ptr(out), int(accumulate), _stream())
    return out


def add_inplace(dst: Tensor, src: Tensor) -> Tensor:
    """dst += src (contiguous fp32 of equal size)."""
    if not (dst.is_contiguous() and src.is_contiguous()) or dst.numel() != src.numel():
        raise ValueError("add_inplace: contiguous operands of equal size expected")
    return col_sum(src.view(1, -1), dst.view(-1), accumulate=True)


def layernorm_bwd(x: Tensor, gamma: Optional[Tensor], dy: Tensor, dx: Optional[Tensor], dgamma: Optional[Tensor] = None,
                  dbeta: Optional[Tensor] = None, eps: float = 1e-5) -> None:
    rows, width, ldx = _f32_2d(x, "ln_bwd.x")
    r2, w2, ldy = _f32_2d(dy, "ln_bwd.dy")
    if (r2, w2) != (rows, width):
        raise ValueError("layernorm_bwd: dy shape mismatch")
    lddx = 0
    if dx is not None:
        r3, w3, lddx = _f32_2d(dx, "ln_bwd.dx")
        if (r3, w3) != (rows, width):
            raise ValueError("layernorm_bwd: dx shape mismatch")
    stats = torch.empty(2 * max(rows, 1), dtype=torch.float32, device=x.device) if dgamma is not None else None
    for t in (gamma, dgamma, dbeta):
        if t is not None:
            _chk(t, torch.float32, "ln_bwd.param", 1)
            if t.numel() != width or not t.is_contiguous():
                raise ValueError("layernorm_bwd: parameter length mismatch")
    _lib.call("otter_layernorm_bwd", _ptr(x), ldx, _ptr(gamma), _ptr(dy), ldy, _ptr(dx), lddx, _ptr(stats), _ptr(dgamma), _ptr(dbeta),
              rows, width, eps, _stream())


def _attn_strides(t: Tensor, name: str, heads: int, hd: int):
    """t: (B, L, >=heads*hd) view with contiguous last dim -> (batch stride, row stride)."""
    _chk(t, torch.float32, name, 3)
    if t.shape[2] != heads * hd:
        raise ValueError(f"{name}: last dim {t.shape[2]} != heads*head_dim {heads * hd}")
    return t.stride(0), t.stride(1)


def small_attn_fwd(q: Tensor, k: Tensor, v: Tensor, heads: int, causal: bool, key_mask: Optional[Tensor] = None,
                   broadcast_q: bool = False) -> Tensor:
    """q (B or 1, Lq, d), k/v (B, Lk, d) fp32 views (same strides for k and v) -> (B, Lq, d)."""
    b, lk, d = k.shape
    hd = d // heads
    lq = q.shape[1]
    sqb, sql = _attn_strides(q, "attn.q", heads, hd)
    skb, skl = _attn_strides(k, "attn.k", heads, hd)
    if _attn_strides(v, "attn.v", heads, hd) != (skb, skl):
        raise ValueError("small_attn: k and v must share strides")
    if broadcast_q:
        sqb = 0
    elif q.shape[0] != b:
        raise ValueError("small_attn: batch mismatch")
    smb = 0
    if key_mask is not None:
        _chk(key_mask, torch.uint8, "attn.mask", 2)
        smb = key_mask.stride(0)
    out = torch.empty(b, lq, d, dtype=torch.float32, device=k.device)
    _lib.call("otter_small_attn_fwd", _ptr(q), sqb, sql, _ptr(k), _ptr(v), skb, skl, _ptr(key_mask), smb, _ptr(out), lq * d, d,
              b, heads, lq, lk, hd, int(causal), _stream())
    return out


def small_attn_bwd(q: Tensor, k: Tensor, v: Tensor, dout: Tensor, dq: Optional[Tensor], dk: Optional[Tensor], dv: Optional[Tensor],
                   heads: int, causal: bool, key_mask: Optional[Tensor] = None, broadcast_q: bool = False) -> None:
    """Accumulates into dq (B, Lq, d) and dk / dv (views with the strides of k)."""
    b, lk, d = k.shape
    hd = d // heads
    lq = q.shape[1]
    sqb, sql = _attn_strides(q, "attn.q", heads, hd)
    skb, skl = _attn_strides(k, "attn.k", heads, hd)
    if broadcast_q:
        sqb = 0
    _chk(dout, torch.float32, "attn.dout", 3)
    if tuple(dout.shape) != (b, lq, d) or not dout.is_contiguous():
        raise ValueError("small_attn_bwd: dout must be contiguous (B, Lq, d)")
    sdqb = sdql = 0
    if dq is not None:
        sdqb, sdql = _attn_strides(dq, "attn.dq", heads, hd)
        if dq.shape[0] != b:
            raise ValueError("small_attn_bwd: dq must have one slab per batch element")
    for t, nm in ((dk, "dk"), (dv, "dv")):
        if t is not None and _attn_strides(t, "attn." + nm, heads, hd) != (skb, skl):
            raise ValueError(f"small_attn_bwd: {nm} must share the strides of k")
    smb = key_mask.stride(0) if key_mask is not None else 0
    _lib.call("otter_small_attn_bwd", _ptr(q), sqb, sql, _ptr(k), _ptr(v), skb, skl, _ptr(key_mask), smb, _ptr(dout), lq * d, d,
              _ptr(dq), sdqb, sdql, _ptr(dk), _ptr(dv), b, heads, lq, lk, hd, int(causal), _stream())


def loss_fwd_bwd(pred: Tensor, gt: Tensor, mse: bool, grad_scale: float, want_grad: bool = True):
    """-> (loss (1,) device tensor, dpred or None)."""
    _chk(pred, torch.float32, "loss.pred")
    _chk(gt, torch.float32, "loss.gt")
    if pred.numel() != gt.numel() or not pred.is_contiguous() or not gt.is_contiguous():
        raise ValueError("loss: pred and gt must be contiguous with the same number of elements")
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    _lib.call("otter_loss_fwd_bwd", _ptr(pred), _ptr(gt), pred.numel(), int(mse), float(grad_scale), _ptr(loss), _ptr(dpred), _stream())
    return loss, dpred


def grad_stats(x: Tensor, workspace: Optional[Tensor] = None) -> Tensor:
    """-> device tensor [sum of squares of finite elements, number of non-finite elements]."""
    _chk(x, torch.float32, "grad_stats.x", 1)
    n_ws = _lib.load().otter_grad_stats_workspace_floats()
    if workspace is None:
        workspace = torch.empty(n_ws, dtype=torch.float32, device=x.device)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    _lib.call("otter_grad_stats", _ptr(x), x.numel(), _ptr(workspace), _ptr(out), _stream())
    return out


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float,
               step: int, inv_scale: float = 1.0, skip_flag: Optional[Tensor] = None) -> None:
    for t in (p, g, m, v):
        _chk(t, torch.float32, "adamw", 1)
        if not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError("adamw_step: flat contiguous buffers of equal length expected")
    _lib.call("otter_adamw_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
              float(weight_decay), int(step), float(inv_scale), _ptr(skip_flag), _stream())


def scale_inplace(x: Tensor, alpha: float) -> None:
    _chk(x, torch.float32, "scale.x")
    if not x.is_contiguous():
        raise ValueError("scale_inplace: contiguous tensor expected")
    _lib.call("otter_scale_inplace", _ptr(x), x.numel(), float(alpha), _stream())


def text_aware_pool_temperature_grad(patches: Tensor, text: Tensor, pos: Tensor, temperature: Tensor, grad_out: Tensor, n_obs: int,
                                     cams: int, t_rep: int, n_patches: Optional[int] = None) -> Tensor:
    """-> fp32 (n_obs, cams): every unit's contribution to d loss / d temperature[cam]."""
    dt = _dt(patches.dtype, "tave_grad.patches", False)
    units, rows, d = patches.shape
    p = n_patches if n_patches is not None else rows
    k = text.shape[1]
    _chk(pos, torch.float32, "tave_grad.pos", 3)
    _chk(grad_out, torch.float32, "tave_grad.grad_out", 4)
    if tuple(pos.shape) != (cams, p, d) or tuple(grad_out.shape) != (cams, n_obs, k, d) or units != n_obs * cams:
        raise ValueError("tave_grad: shape mismatch")
    for t in (patches, text, pos, temperature, grad_out):
        if not t.is_contiguous():
            raise ValueError("tave_grad: operands must be contiguous")
    partial = torch.empty(n_obs, cams, dtype=torch.float32, device=patches.device)
    base = ctypes.c_void_p(patches.data_ptr() + (rows - p) * d * 2)
    _lib.call("otter_text_aware_pool_temperature_grad", dt, base, rows * d, _ptr(text), _ptr(pos), _ptr(temperature), _ptr(grad_out),
              _ptr(partial), n_obs, cams, p, d, k, t_rep, _stream())
    return partial
