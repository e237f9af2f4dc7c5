"""Device-side AttentionPooling (``otter/policy/transformer.py:162-232``).

The only part with real FLOPs -- projecting the (n_obs*nk) key/value tokens from D_in to d for every block --
runs as ONE fp32 MFMA GEMM: the tokens are normalised once (no affine) and each block's ``kv_layer_norm``
affine is folded into its projection weights (W' = W diag(gamma), b' = W beta + b), the K and V weights of all
blocks are stacked along N.  The query side runs in the fp32 readout kernel.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .torch_ops import kernels as ops

Tensor = torch.Tensor


class PoolingWeights:
    def __init__(self, sd: Dict[str, Tensor], prefix: str, heads: int):
        q = sd[prefix + "query"]
        self.device = q.device
        _, self.readouts, self.d = q.shape
        self.heads = heads
        if self.d % heads != 0:
            raise ValueError(f"{prefix}: readout width {self.d} not divisible by {heads} heads")
        self.query = q.reshape(self.readouts, self.d).float().contiguous()
        self.final_w = sd[prefix + "layer_norm.weight"].float().contiguous()
        self.final_b = sd[prefix + "layer_norm.bias"].float().contiguous()
        n = 0
        while f"{prefix}blocks.{n}.q_layer_norm.weight" in sd:
            n += 1
        self.n_blocks = n
        d = self.d
        ws, bs = [], []
        self.block_params: List[Tensor] = []              # 12 tensors per block, in the order of otter_readout_block's fields
        for i in range(n):
            p = f"{prefix}blocks.{i}."
            gamma, beta = sd[p + "kv_layer_norm.weight"].float(), sd[p + "kv_layer_norm.bias"].float()
            bias = sd[p + "attention.in_proj_bias"].float()
            for name, sl in (("k", slice(d, 2 * d)), ("v", slice(2 * d, 3 * d))):
                w = sd[p + f"attention.{name}_proj_weight"].float()
                ws.append(w * gamma[None, :])
                bs.append(w @ beta + bias[sl])
            fields = dict(
                q_ln_w=sd[p + "q_layer_norm.weight"], q_ln_b=sd[p + "q_layer_norm.bias"],
                wq_t=sd[p + "attention.q_proj_weight"].t(), bq=bias[:d],
                wo_t=sd[p + "attention.out_proj.weight"].t(), bo=sd[p + "attention.out_proj.bias"],
                ln_w=sd[p + "layer_norm.weight"], ln_b=sd[p + "layer_norm.bias"],
                fc1_t=sd[p + "mlp.fc1.weight"].t(), fc1_b=sd[p + "mlp.fc1.bias"],
                fc2_t=sd[p + "mlp.fc2.weight"].t(), fc2_b=sd[p + "mlp.fc2.bias"])
            assert tuple(fields) == ops.READOUT_FIELDS
            self.block_params += [t.float().contiguous() for t in fields.values()]
        self.hidden = sd[prefix + "blocks.0.mlp.fc1.weight"].shape[0]
        self.d_in = ws[0].shape[1]
        self.kv_n = 2 * d * n
        self.kv_w = ops.pack_weight(torch.cat(ws, 0), torch.float32)
        self.kv_b = torch.cat(bs, 0).contiguous()

    def problem(self, x: Tensor, mask: Optional[Tensor], out: Tensor, n_obs: int, nk: int):
        """Key / value projection of ``x`` (launched here) and the argument tuple of the readout that consumes it
        (``readout_pool_multi``: the poolings of one encoder call share a launch)."""
        xn = ops.layernorm(x, None, None, out_dtype=torch.float32)
        kv = ops.gemm(xn, self.kv_w, self.kv_n, self.kv_b, out_f32=True)
        m8 = None
        if mask is not None:
            m8 = mask.to(torch.uint8).contiguous()
        return (self.block_params, self.query, self.final_w, self.final_b, kv, m8, out, n_obs, nk, self.readouts, self.d, self.heads, self.hidden)

    def forward(self, x: Tensor, mask: Optional[Tensor], out: Tensor, n_obs: int, nk: int) -> Tensor:
        """x fp32 (n_obs*nk, D_in); mask bool/uint8 (n_obs, nk) with True = ignore; out fp32 view (n_obs, readouts*d)."""
        ops.readout_pool_multi([self.problem(x, mask, out, n_obs, nk)])
        return out
