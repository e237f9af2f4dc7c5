"""The two single-GPU configurations at their REAL sizes (default ViT-L/14 model): the batch-64 rollout with history 12
across the window boundary, and one data-parallel rank's training step (16 windows of 12 observations).  The oracle
covers what it can finish in seconds (two episodes / one window); batch independence covers the rest."""
import numpy as np
import pytest
import torch

import golden_inputs as GI
from oracle import otter_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from otter_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def default_model(dev):
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.otter import OTTER
    from otter_amd.params import otter_specs, random_state_dict
    mc, sc = ModelConfig(), SharedConfig()
    sd = random_state_dict(otter_specs(mc, sc), seed=0)
    model = OTTER(mc, sc)
    model.load_state_dict(sd, strict=True)
    return model.to(dev), sd, mc, sc


def test_batch64_history12_rollout_at_full_size(dev, default_model):
    """64 concurrent episodes x ViT-L/14, 14 steps = 12 filling the K/V cache + 2 after the window starts sliding.
    Episodes 0 and 63 against single-episode rollouts of the fp32 CPU oracle (1e-3 on actions); episodes 7 and 31
    against the same episodes run alone on the device (batch independence up to the split-K summation order)."""
    from otter_amd.engine import BatchedRollout
    model, sd, mc, sc = default_model
    model.eval()
    d = model.device_model()
    B, steps = 64, sc.seq_length + 2
    rng = np.random.default_rng(17)
    text = GI._tokens(rng, B, (4, 6, 9))
    tf = d.clip.text_tokens_cached(text)
    proprio = torch.from_numpy(rng.standard_normal((steps, B, 10)).astype(np.float32))
    g = torch.Generator().manual_seed(17)
    batch = BatchedRollout(d, B)
    oracles = {e: R.Rollout(sd, R.Dims()) for e in (0, 63)}
    alone = {e: BatchedRollout(d, 1) for e in (7, 31)}
    worst_oracle = worst_alone = 0.0
    for t in range(steps):
        images = torch.randint(0, 256, (B, 2, 3, 224, 224), dtype=torch.uint8, generator=g)       # 19 MB per step
        got = batch.step(images.view(B * 2, 3, 224, 224).to(dev), tf, proprio[t].to(dev)).cpu()
        assert got.shape == (B, sc.action_horizon, mc.action_dim) and torch.isfinite(got).all()
        for e, ro in oracles.items():
            ref = ro.step(R.clip_transform(images[e]), text[e], proprio[t, e])
            worst_oracle = max(worst_oracle, (got[e] - ref).abs().max().item())
        for e, ro in alone.items():
            a = ro.step(images[e].to(dev), tf[e:e + 1], proprio[t, e:e + 1].to(dev)).cpu()[0]
            worst_alone = max(worst_alone, (got[e] - a).abs().max().item())
    assert list(batch.lens_host) == [sc.seq_length] * B
    print(f"batch 64 x history 12, {steps} steps: max |action - oracle| = {worst_oracle:.2e}, max |batch - alone| = {worst_alone:.2e}")
    assert worst_oracle <= 1e-3 and worst_alone <= 5e-4


def test_config5_default_model_window32(dev):
    """Long-horizon configuration at the REAL model size: default ViT-L/14 + 768-wide 8-layer policy, history window 32,
    64 concurrent episodes (one GPU's share of 512), 34 steps = 32 filling the K/V cache ([64][32][2304] fp32 per layer)
    + 2 after the window starts sliding (full-window recompute: 2048 rows through every policy GEMM).  Episodes 0 and 63
    against single-episode rollouts of the fp32 CPU oracle (1e-3 on actions), episodes 7 and 31 against the same episodes
    run alone on the device."""
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.engine import BatchedRollout
    from otter_amd.otter import OTTER
    from otter_amd.params import otter_specs, random_state_dict
    mc, sc = ModelConfig(), SharedConfig(seq_length=32)
    sd = random_state_dict(otter_specs(mc, sc), seed=5)
    assert tuple(sd["pe.pe"].shape) == (1, 32, mc.transformer_dim)
    model = OTTER(mc, sc)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    d = model.device_model()
    B, steps = 64, sc.seq_length + 2
    rng = np.random.default_rng(29)
    text = GI._tokens(rng, B, (4, 6, 9))
    tf = d.clip.text_tokens_cached(text)
    proprio = torch.from_numpy(rng.standard_normal((steps, B, 10)).astype(np.float32))
    g = torch.Generator().manual_seed(29)
    batch = BatchedRollout(d, B)
    dims = R.Dims(seq_length=sc.seq_length)
    oracles = {e: R.Rollout(sd, dims) for e in (0, 63)}
    alone = {e: BatchedRollout(d, 1) for e in (7, 31)}
    worst_oracle = worst_alone = 0.0
    worst_slide = 0.0
    for t in range(steps):
        images = torch.randint(0, 256, (B, 2, 3, 224, 224), dtype=torch.uint8, generator=g)
        got = batch.step(images.view(B * 2, 3, 224, 224).to(dev), tf, proprio[t].to(dev)).cpu()
        assert got.shape == (B, sc.action_horizon, mc.action_dim) and torch.isfinite(got).all()
        for e, ro in oracles.items():
            ref = ro.step(R.clip_transform(images[e]), text[e], proprio[t, e])
            err = (got[e] - ref).abs().max().item()
            worst_oracle = max(worst_oracle, err)
            if t >= sc.seq_length:
                worst_slide = max(worst_slide, err)
        for e, ro in alone.items():
            a = ro.step(images[e].to(dev), tf[e:e + 1], proprio[t, e:e + 1].to(dev)).cpu()[0]
            worst_alone = max(worst_alone, (got[e] - a).abs().max().item())
    assert list(batch.lens_host) == [sc.seq_length] * B
    print(f"batch 64 x history 32 (default model), {steps} steps: max |action - oracle| = {worst_oracle:.2e} "
          f"(sliding steps {worst_slide:.2e}), max |batch - alone| = {worst_alone:.2e}")
    assert worst_oracle <= 1e-3 and worst_alone <= 5e-4


def test_training_step_at_one_ranks_full_size(dev, default_model):
    """One data-parallel rank of the global-batch-128 training step: 16 windows x 12 observations x 2 cameras = 384
    ViT-L/14 images.  Actions and loss of window 0 against the fp32 CPU oracle; loss + backward over the whole rank
    batch gives a finite gradient for every trainable tensor."""
    model, sd, mc, sc = default_model
    model.train()
    b, t = 16, sc.seq_length
    rng = np.random.default_rng(23)
    g = torch.Generator().manual_seed(23)
    images = torch.randint(0, 256, (b, t, 2, 3, 224, 224), dtype=torch.uint8, generator=g)
    text = GI._tokens(rng, b, (4, 6, 9))
    proprio = torch.from_numpy(rng.standard_normal((b, t, 10)).astype(np.float32))
    gt = torch.from_numpy(rng.standard_normal((b, t, sc.action_horizon, mc.action_dim)).astype(np.float32))
    with torch.no_grad():
        actions = model(images.to(dev), text, proprio.to(dev)).cpu()
        ref_actions = R.forward(images[:1], text[:1], proprio[:1], sd, R.Dims())
        ref_loss = torch.nn.functional.l1_loss(ref_actions, gt[:1]).item()
        loss0 = model(images[:1].to(dev), text[:1], proprio[:1].to(dev), gt[:1].to(dev)).item()
    err = (actions[:1] - ref_actions).abs().max().item()
    print(f"training batch 16 x 12: window 0 max |action - oracle| = {err:.2e}, loss {loss0:.6f} vs oracle {ref_loss:.6f}")
    assert actions.shape == (b, t, sc.action_horizon, mc.action_dim) and err <= 1e-3 and abs(loss0 - ref_loss) <= 1e-3
    for p in model.parameters():
        p.grad = None
    loss = model(images.to(dev), text, proprio.to(dev), gt.to(dev))
    want = torch.nn.functional.l1_loss(actions, gt).item()             # the loss of the device's own actions, on the host
    assert abs(loss.item() - want) <= 1e-4
    loss.backward()
    n = 0
    for name, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
            n += p.numel()
            assert name.startswith("clip_model.") is False
    assert n == 31_055_290
    assert float(model.action_head.mlp[0].weight.grad.norm()) > 0 and float(model.visual_extraction[0].temperature.grad.abs()) > 0
