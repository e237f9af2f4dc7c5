"""Seeded inputs shared by ``tools/gen_golden.py`` (which runs the reference on them) and the tests.

All randomness comes from ``numpy.random.default_rng(seed)`` (stream-stable across numpy versions)
so the inputs behind ``tests/golden/*.npz`` can be rebuilt anywhere without storing them.
"""
from __future__ import annotations

import numpy as np
import torch

from otter_amd.config import ModelConfig, SharedConfig


def _randn(seed, *shape):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32))


def _unit(x):
    return x / x.norm(dim=-1, keepdim=True)


# ---------------------------------------------------------------- text-aware visual extraction
TAVE_CASES = [
    dict(name="tiny", M=2, P=16, K=3, D=8, temperature=0.07, seed=11),
    dict(name="full", M=1, P=256, K=15, D=768, temperature=0.07, seed=12),
    dict(name="full_b3", M=3, P=256, K=15, D=768, temperature=0.07, seed=13),
    dict(name="temp_clamped_high", M=2, P=64, K=15, D=128, temperature=150.0, seed=14),
    dict(name="temp_small", M=2, P=64, K=15, D=128, temperature=0.005, seed=15),
]


def tave_inputs(case):
    p = _unit(_randn(case["seed"], case["M"], case["P"], case["D"]))
    t = _unit(_randn(case["seed"] + 1000, case["M"], case["K"], case["D"]))
    return p, t


# ---------------------------------------------------------------- attention pooling
POOL_CASES = [
    dict(name="vision_nomask", D_in=768, D_out=256, heads=8, layers=2, readouts=4, B=3, K=15, mask=False, seed=21),
    dict(name="text_nomask", D_in=768, D_out=128, heads=8, layers=2, readouts=4, B=3, K=15, mask=False, seed=22),
    dict(name="vision_mask", D_in=768, D_out=256, heads=8, layers=2, readouts=4, B=3, K=15, mask=True, seed=23),
    dict(name="text_mask", D_in=768, D_out=128, heads=8, layers=2, readouts=4, B=3, K=15, mask=True, seed=24),
    dict(name="small", D_in=128, D_out=64, heads=4, layers=1, readouts=2, B=2, K=7, mask=False, seed=25),
]


def pool_inputs(case):
    x = _randn(case["seed"] + 500, case["B"], case["K"], case["D_in"])
    mask = None
    if case["mask"]:
        rng = np.random.default_rng(case["seed"] + 501)
        m = rng.random((case["B"], case["K"])) < 0.4
        m[:, 1] = False                       # at least one key stays visible per row
        mask = torch.from_numpy(m)
    return x, mask


# ---------------------------------------------------------------- policy transformer
POLICY_CASES = [
    dict(name="tiny", dim=64, heads=2, layers=2, expansion=1, B=2, T=6, seed=31),
    dict(name="tiny_exp4", dim=64, heads=4, layers=1, expansion=4, B=1, T=4, seed=32),
    dict(name="full_T12", dim=768, heads=8, layers=8, expansion=1, B=1, T=12, seed=33),
]


def policy_inputs(case):
    return _randn(case["seed"] + 700, case["B"], case["T"], case["dim"])


HEAD_CASES = [dict(name="head_full", dim=768, A=10, H=12, B=5, seed=41),
              dict(name="head_small", dim=32, A=7, H=3, B=2, seed=42)]


def head_inputs(case):
    return _randn(case["seed"] + 800, case["B"], case["dim"])


PROPRIO_CASES = [dict(name="proprio_full", inp=10, hidden=256, out=64, B=5, seed=51)]


def proprio_inputs(case):
    return _randn(case["seed"] + 900, case["B"], case["inp"])


# ---------------------------------------------------------------- host helpers
def text_tokens():
    """Three token rows: a short prompt, a prompt longer than first_k, and an empty prompt."""
    t = torch.zeros(3, 77, dtype=torch.int32)
    rng = np.random.default_rng(61)
    for row, n in enumerate((5, 30, 0)):
        t[row, 0] = 49406
        t[row, 1:1 + n] = torch.from_numpy(rng.integers(1, 49406, n).astype(np.int32))
        t[row, 1 + n] = 49407
    return t


def transform_inputs():
    rng = np.random.default_rng(62)
    u8 = torch.from_numpy(rng.integers(0, 256, (2, 2, 3, 8, 8), dtype=np.uint8))
    f_big = torch.from_numpy((rng.random((2, 3, 8, 8)) * 255).astype(np.float32))
    f_unit = torch.from_numpy(rng.random((2, 3, 8, 8)).astype(np.float32))
    return u8, f_big, f_unit


def _rand_rot6d(rng, n):
    from scipy.spatial.transform import Rotation
    m = Rotation.from_rotvec(rng.standard_normal((n, 3))).as_matrix()
    return m[:, :2, :].reshape(n, 6)


def abs_action_inputs():
    """(action (S,T,10), proprio (S,1,10)) pairs: orthonormal deltas, a non-orthonormal delta
    (exercises the Gram-Schmidt branch), and a proprio given as 2-D."""
    rng = np.random.default_rng(63)
    out = []
    S, T = 2, 12
    act = np.concatenate([rng.standard_normal((S * T, 3)) * 0.05, _rand_rot6d(rng, S * T),
                          rng.random((S * T, 1))], axis=-1).reshape(S, T, 10)
    prop = np.concatenate([rng.standard_normal((S, 3)), _rand_rot6d(rng, S), rng.random((S, 1))], axis=-1).reshape(S, 1, 10)
    out.append((act, prop))
    act2 = act.copy()
    act2[:, :, 3:9] += rng.standard_normal((S, T, 6)) * 0.05        # network-like output: not orthonormal
    out.append((act2, prop))
    out.append((act2[:1], prop[0]))                                  # 2-D proprio (S, D)
    return out


def action10_inputs():
    rng = np.random.default_rng(64)
    n = 4
    return np.concatenate([rng.standard_normal((n, 3)), _rand_rot6d(rng, n), rng.random((n, 1))], axis=-1)


CROP_SIZES = [(64, 64), (80, 48), (48, 80), (81, 48), (48, 81)]


# ---------------------------------------------------------------- end-to-end (tiny CLIP)
E2E_CASES = [
    dict(name="tiny_default", pool_true_text=False, mse=False, seq_length=4, B=2, T=3, rollout_steps=7, seed=71),
    dict(name="tiny_masked_mse", pool_true_text=True, mse=True, seq_length=3, B=1, T=3, rollout_steps=5, seed=72),
]


def e2e_configs(case):
    mc = ModelConfig(clip_model="tiny-test", pool_true_text=case["pool_true_text"],
                     transformer_dim=128, transformer_layers=2, transformer_heads=2,
                     vision_pooling_output_dim=128, text_pooling_output_dim=64, proprio_hidden_dim=64,
                     proprio_output_dim=32, pooling_heads=4)
    sc = SharedConfig(image_size=112, seq_length=case["seq_length"], mse_loss=case["mse"], action_horizon=4)
    return mc, sc


def _tokens(rng, b, lengths):
    t = torch.zeros(b, 77, dtype=torch.int32)
    for row in range(b):
        n = lengths[row % len(lengths)]
        t[row, 0] = 49406
        t[row, 1:1 + n] = torch.from_numpy(rng.integers(1, 49406, n).astype(np.int32))
        t[row, 1 + n] = 49407
    return t


def e2e_inputs(case):
    rng = np.random.default_rng(case["seed"])
    B, T = case["B"], case["T"]
    images = torch.from_numpy(rng.integers(0, 256, (B, T, 2, 3, 112, 112), dtype=np.uint8))
    text = _tokens(rng, B, (5, 9))
    proprio = torch.from_numpy(rng.standard_normal((B, T, 10)).astype(np.float32))
    gt = torch.from_numpy(rng.standard_normal((B, T, 4, 10)).astype(np.float32))
    return images, text, proprio, gt


def rollout_inputs(case):
    rng = np.random.default_rng(case["seed"] + 1)
    n = case["rollout_steps"]
    images = torch.from_numpy(rng.integers(0, 256, (n, 2, 3, 112, 112), dtype=np.uint8))
    text = _tokens(rng, 1, (6,))[0]
    proprio = torch.from_numpy(rng.standard_normal((n, 10)).astype(np.float32))
    return images, text, proprio


def sd_probe(sd):
    """A few checksums of a generated state_dict, to detect a drifting weight generator."""
    keys = ["clip_model.visual.conv1.weight", "clip_model.token_embedding.weight",
            "policy.layers.0.attention.query.weight", "action_head.mlp.2.bias", "text_pooling.query"]
    return np.asarray([float(sd[k].double().abs().sum()) for k in keys])
