"""Whole-path parity on the GPU: the HIP OTTER model against (a) the fixtures produced by the reference's own
OTTER class (tiny CLIP) and (b) the fp32 CPU oracle at the full default configuration (ViT-L/14).

Tolerances come from the task statement: visual tokens cosine >= 0.9999, actions within 1e-3 abs of fp32.
"""
import numpy as np
import pytest
import torch

import golden_inputs as GI
from oracle import clip_ref
from oracle import otter_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from otter_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _model(mc, sc, seed, dev):
    from otter_amd.otter import OTTER
    from otter_amd.params import otter_specs, random_state_dict
    sd = random_state_dict(otter_specs(mc, sc), seed=seed)
    model = OTTER(mc, sc)
    model.load_state_dict(sd, strict=True)
    return model.to(dev).eval(), sd


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()))


@pytest.mark.parametrize("case", GI.E2E_CASES, ids=lambda c: c["name"])
def test_tiny_model_matches_reference_fixtures(dev, golden, case):
    g = golden("otter_e2e")
    n = case["name"]
    mc, sc = GI.e2e_configs(case)
    model, sd = _model(mc, sc, case["seed"], dev)
    np.testing.assert_allclose(GI.sd_probe(sd), g[n + "__sd_probe"], rtol=1e-12)
    images, text, proprio, gt = GI.e2e_inputs(case)

    pf, tf = model.extract_clip_features(images.to(dev), text)
    ref_pf = torch.from_numpy(g[n + "__patch_feat_cam0_first"])
    assert _cos(pf[0][0, :4].cpu(), ref_pf) >= 0.9999
    assert _cos(tf[0].cpu(), torch.from_numpy(g[n + "__text_feat_first"])) >= 0.9999

    enc = model.forward_encoder(images.to(dev), text, proprio.to(dev),
                                R.create_text_mask(text, 49406, 49407, mc.first_k_tokens) if mc.pool_true_text else None)
    ref_enc = torch.from_numpy(g[n + "__encoder"])
    assert _cos(enc.cpu(), ref_enc) >= 0.9999

    actions = model(images.to(dev), text, proprio.to(dev)).cpu()
    ref_actions = torch.from_numpy(g[n + "__actions"])
    assert actions.shape == ref_actions.shape
    err = (actions - ref_actions).abs().max().item()
    print(f"{n}: max |action - ref| = {err:.2e}, |ref| max = {ref_actions.abs().max():.3f}")
    assert err <= 1e-3

    loss = model(images.to(dev), text, proprio.to(dev), gt.to(dev)).item()
    assert abs(loss - float(g[n + "__loss"])) <= 1e-3

    # rollout through the window boundary: cached steps while the window fills, recompute after it slides
    r_images, r_text, r_proprio = GI.rollout_inputs(case)
    mask = R.create_text_mask(r_text, 49406, 49407, mc.first_k_tokens) if mc.pool_true_text else None
    model.reset_cache()
    ref_steps = g[n + "__rollout"]
    worst = 0.0
    for t in range(r_images.shape[0]):
        a = model.forward_inference(r_images[t].to(dev), r_text, r_proprio[t].to(dev), mask).cpu().numpy()
        assert model.cache_size == min(t + 1, sc.seq_length)
        assert model.cached_transformer_input.shape[1] == model.cache_size
        worst = max(worst, float(np.abs(a - ref_steps[t]).max()))
    print(f"{n}: rollout max |action - ref| = {worst:.2e}")
    assert worst <= 1e-3
    model.reset_cache()
    assert model.cache_size == 0 and model.cached_transformer_input is None
    a0 = model.forward_inference(r_images[0].to(dev), r_text, r_proprio[0].to(dev), mask).cpu().numpy()
    assert float(np.abs(a0 - ref_steps[0]).max()) <= 1e-3


def test_default_config_vit_l14_matches_oracle(dev):
    """Random-init default configuration (the reference's tests/test_dummy_model.py set-up), batch 1, two 224^2
    cameras, one prompt: visual tokens, text tokens, policy input and actions against the fp32 CPU oracle."""
    from otter_amd.config import ModelConfig, SharedConfig
    mc, sc = ModelConfig(), SharedConfig()
    model, sd = _model(mc, sc, 0, dev)
    dims = R.Dims()
    rng = np.random.default_rng(5)
    steps = 3
    images = torch.from_numpy(rng.integers(0, 256, (steps, 2, 3, 224, 224), dtype=np.uint8))
    text = torch.zeros(77, dtype=torch.int32)
    ids = rng.integers(1, 49406, 6)
    text[0], text[1:7], text[7] = 49406, torch.from_numpy(ids.astype(np.int32)), 49407
    proprio = torch.from_numpy(rng.standard_normal((steps, 10)).astype(np.float32))

    # visual / text tokens of the first observation
    pf, tf = model.extract_clip_features(images[:1, None].to(dev), text[None])
    with torch.no_grad():
        ref_pf = clip_ref.patch_features(R.clip_transform(images[0]), sd)
        ref_tf = clip_ref.text_features(text[None], 15, sd)
    cos_patch = min(_cos(pf[c][0].cpu(), ref_pf[c]) for c in range(2))
    worst_tok = min(_cos(pf[c][0, i].cpu(), ref_pf[c, i]) for c in range(2) for i in range(0, 256, 17))
    cos_text = _cos(tf[0].cpu(), ref_tf[0])
    print(f"ViT-L/14: patch cosine {cos_patch:.6f} (worst sampled token {worst_tok:.6f}), text cosine {cos_text:.6f}")
    assert cos_patch >= 0.9999 and worst_tok >= 0.9999 and cos_text >= 0.9999

    ro = R.Rollout(sd, dims)
    model.reset_cache()
    worst = 0.0
    for t in range(steps):
        ref = ro.step(R.clip_transform(images[t]), text, proprio[t]).numpy()
        got = model.forward_inference(images[t].to(dev), text, proprio[t].to(dev)).cpu().numpy()
        worst = max(worst, float(np.abs(got - ref).max()))
        print(f"step {t}: max |action - oracle| = {np.abs(got - ref).max():.2e}  (|oracle| max {np.abs(ref).max():.3f})")
    assert worst <= 1e-3


@pytest.mark.parametrize("dtype,bound", [(torch.float16, 1e-3), (torch.bfloat16, 1e-3)])
def test_outlier_channel_weights_vit_l14(dev, dtype, bound):
    """Trained CLIP towers carry a few residual-stream channels tens of times larger than the rest, read through small
    LayerNorm gains; N(0, sigma) weights never produce that.  Default configuration with 6 such channels per tower (x60):
    the 16-bit residual stream and the LayerNorm folded into 16-bit weights against the fp32 oracle on the same weights.
    (Parity for the PUBLISHED weights themselves stays unpinned: they are not available offline.)"""
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.otter import OTTER
    from otter_amd.params import otter_specs, random_state_dict
    mc, sc = ModelConfig(), SharedConfig()
    sd = random_state_dict(otter_specs(mc, sc), seed=0, outlier_channels=6, outlier_scale=60.0)
    model = OTTER(mc, sc)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    model.clip_compute_dtype = dtype
    rng = np.random.default_rng(6)
    images = torch.from_numpy(rng.integers(0, 256, (2, 3, 224, 224), dtype=np.uint8))
    text = torch.zeros(77, dtype=torch.int32)
    text[0], text[1:7], text[7] = 49406, torch.from_numpy(rng.integers(1, 49406, 6).astype(np.int32)), 49407
    proprio = torch.from_numpy(rng.standard_normal(10).astype(np.float32))
    clip = model.clip_device()
    clip.debug_stream = True
    pf, tf = model.extract_clip_features(images[None, None].to(dev), text[None])
    clip.debug_stream = False
    with torch.no_grad():
        ref_pf = clip_ref.patch_features(R.clip_transform(images), sd)
        ref_tf = clip_ref.text_features(text[None], 15, sd)
    cos_patch = min(_cos(pf[c][0].cpu(), ref_pf[c]) for c in range(2))
    cos_text = _cos(tf[0].cpu(), ref_tf[0])
    ref = R.Rollout(sd, R.Dims()).step(R.clip_transform(images), text, proprio).numpy()
    got = model.forward_inference(images.to(dev), text, proprio.to(dev)).cpu().numpy()
    err = float(np.abs(got - ref).max())
    peak = max(clip.stream_absmax)
    print(f"outlier weights, {dtype}: residual stream peaks at {peak:.0f} (typical element ~1), patch cosine {cos_patch:.6f}, "
          f"text cosine {cos_text:.6f}, max |action - oracle| = {err:.2e}")
    assert peak > 100 and peak < 65504                 # the stream really carries outliers, and stays inside fp16's range
    assert cos_patch >= 0.9999 and cos_text >= 0.9999 and err <= bound


@pytest.mark.parametrize("clip_model,res", [("ViT-B/16", 224), ("ViT-B/32", 224), ("ViT-L/14@336px", 336)])
def test_other_clip_backbones_match_oracle(dev, clip_model, res):
    """The other CLIP ViTs the reference's config accepts: 196 / 49 / 576 patches (the first two are not whole 16-patch
    tiles for the pooling kernel), 197 / 50 / 577 tokens per sequence, embedding widths 512 / 768.  Visual / text tokens
    and one rollout step vs the oracle."""
    from otter_amd.config import ModelConfig, SharedConfig
    mc, sc = ModelConfig(clip_model=clip_model), SharedConfig(image_size=res)
    model, sd = _model(mc, sc, 3, dev)
    rng = np.random.default_rng(11)
    images = torch.from_numpy(rng.integers(0, 256, (1, 2, 3, res, res), dtype=np.uint8))
    text = torch.zeros(77, dtype=torch.int32)
    text[0], text[1:5], text[5] = 49406, torch.from_numpy(rng.integers(1, 49406, 4).astype(np.int32)), 49407
    proprio = torch.from_numpy(rng.standard_normal((1, 10)).astype(np.float32))
    pf, tf = model.extract_clip_features(images[:1, None].to(dev), text[None])
    with torch.no_grad():
        ref_pf = clip_ref.patch_features(R.clip_transform(images[0]), sd)
        ref_tf = clip_ref.text_features(text[None], 15, sd)
    cos_patch = min(_cos(pf[c][0].cpu(), ref_pf[c]) for c in range(2))
    cos_text = _cos(tf[0].cpu(), ref_tf[0])
    ref = R.Rollout(sd, R.Dims()).step(R.clip_transform(images[0]), text, proprio[0]).numpy()
    got = model.forward_inference(images[0].to(dev), text, proprio[0].to(dev)).cpu().numpy()
    err = float(np.abs(got - ref).max())
    print(f"{clip_model}: {pf[0].shape[1]} patches, patch cosine {cos_patch:.6f}, text cosine {cos_text:.6f}, max |action - oracle| = {err:.2e}")
    assert cos_patch >= 0.9999 and cos_text >= 0.9999 and err <= 1e-3


def test_batched_rollout_equals_single_episode(dev):
    """B concurrent episodes give each episode what it gets alone; a per-episode reset does not disturb the others."""
    from otter_amd.engine import BatchedRollout
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, sd = _model(mc, sc, case["seed"], dev)
    d = model.device_model()
    B, steps = 3, sc.seq_length + 3
    rng = np.random.default_rng(9)
    images = torch.from_numpy(rng.integers(0, 256, (steps, B, 2, 3, 112, 112), dtype=np.uint8)).to(dev)
    proprio = torch.from_numpy(rng.standard_normal((steps, B, 10)).astype(np.float32)).to(dev)
    text = GI._tokens(rng, B, (4, 7, 9))
    tf = d.clip.text_tokens_cached(text)
    batch = BatchedRollout(d, B)
    got = [batch.step(images[t].reshape(B * 2, 3, 112, 112), tf, proprio[t]).cpu() for t in range(steps)]
    for e in range(B):
        single = BatchedRollout(d, 1)
        for t in range(steps):
            a = single.step(images[t, e], tf[e:e + 1], proprio[t, e:e + 1]).cpu()
            torch.testing.assert_close(a[0], got[t][e], rtol=1e-3, atol=2e-4)
    # reset episode 1 only: it restarts, the others continue
    batch.reset([1])
    nxt = batch.step(images[0].reshape(B * 2, 3, 112, 112), tf, proprio[0]).cpu()
    fresh = BatchedRollout(d, 1)
    torch.testing.assert_close(nxt[1], fresh.step(images[0, 1], tf[1:2], proprio[0, 1:2]).cpu()[0], rtol=1e-3, atol=2e-4)
    assert list(batch.lens_host) == [sc.seq_length, 1, sc.seq_length]


def test_staggered_episodes_keep_their_kv_cache(dev):
    """A server whose episodes start and end at different times (window 8, five episodes, a schedule of partial resets):
    every episode gets, at every step, what the fp32 CPU oracle gives that episode alone; the engine's mode per step is what
    the per-episode rule says -- cached while the episode's own window is filling (at its own position), recompute once it
    slides -- so a partial reset never costs the other episodes their K/V cache."""
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.engine import BatchedRollout
    mc = ModelConfig(clip_model="tiny-test", transformer_dim=128, transformer_layers=2, transformer_heads=2, vision_pooling_output_dim=128,
                     text_pooling_output_dim=64, proprio_hidden_dim=64, proprio_output_dim=32, pooling_heads=4)
    sc = SharedConfig(image_size=112, seq_length=8, action_horizon=4)
    model, sd = _model(mc, sc, 31, dev)
    d = model.device_model()
    B, steps = 5, 22
    resets = {2: [1], 3: [3, 4], 5: [0], 9: [2], 13: [1, 4], 17: [0, 1, 2, 3, 4], 19: [2]}      # before the step with that number
    rng = np.random.default_rng(19)
    images = torch.from_numpy(rng.integers(0, 256, (steps, B, 2, 3, 112, 112), dtype=np.uint8))
    proprio = torch.from_numpy(rng.standard_normal((steps, B, 10)).astype(np.float32))
    text = GI._tokens(rng, B, (4, 7, 9))
    tf = d.clip.text_tokens_cached(text)
    dims = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                  action_dim=mc.action_dim, action_horizon=sc.action_horizon, seq_length=sc.seq_length)
    batch = BatchedRollout(d, B)
    oracles = [R.Rollout(sd, dims) for _ in range(B)]
    lens = np.zeros(B, dtype=int)
    worst, modes = 0.0, []
    for t in range(steps):
        for e in resets.get(t, []):
            oracles[e].reset()
            lens[e] = 0
        if t in resets:
            batch.reset(resets[t])
        got = batch.step(images[t].reshape(B * 2, 3, 112, 112).to(dev), tf, proprio[t].to(dev)).cpu()
        filling = lens < sc.seq_length
        want_mode = "cached" if filling.all() else ("recompute" if not filling.any() else "mixed")
        assert batch.last_mode == want_mode, (t, batch.last_mode, want_mode, lens.tolist())
        modes.append(batch.last_mode)
        lens = np.minimum(lens + 1, sc.seq_length)
        assert list(batch.lens_host) == lens.tolist()
        for e in range(B):
            ref = oracles[e].step(R.clip_transform(images[t, e]), text[e], proprio[t, e])
            worst = max(worst, (got[e] - ref).abs().max().item())
    print(f"staggered episodes: modes {''.join(m[0] for m in modes)}, max |action - oracle| = {worst:.2e}")
    assert {"cached", "mixed"} <= set(modes) and worst <= 1e-3
    with pytest.raises(IndexError):
        batch.reset([B])


def test_openclip_gelu_backbone_text_aware_extraction(dev):
    """OpenCLIP ViT-L/14 variant (GELU instead of QuickGELU; script/open_clip_visualization.py:38-117): per (image, prompt)
    pair, 15 text-token queries plus the pooled text feature as a 16th query, softmax temperature 0.005.
    Four pairs against the fp32 CPU oracle, then the same pairs inside a batch of 256 (one GPU's share of the batch-256
    extraction): results do not depend on the batch they are computed in."""
    from otter_amd import ops
    from otter_amd.config import ModelConfig, SharedConfig, clip_arch
    from otter_amd.device_model import CLIP_MEAN, CLIP_STD, ClipDevice
    from otter_amd.params import otter_specs, random_state_dict, sincos_position_embedding
    mc, sc = ModelConfig(clip_model="open_clip:ViT-L-14"), SharedConfig()
    arch = clip_arch(mc.clip_model)
    assert arch.activation == "gelu"
    sd = random_state_dict(otter_specs(mc, sc), seed=3)
    clip = ClipDevice({k: v for k, v in sd.items() if k.startswith("clip_model.")}, arch, 15, dev)
    rng = np.random.default_rng(11)
    n_big, n_small = 256, 4
    images = torch.from_numpy(rng.integers(0, 256, (n_big, 3, 224, 224), dtype=np.uint8))
    text = GI._tokens(rng, n_big, (3, 5, 8, 11, 17))                       # the 17-token prompt ends past the first 15 positions
    pos = sincos_position_embedding(256, 768)
    temp = torch.tensor([0.005])
    pos_t = ops.pos_emb_pack(pos.to(dev), torch.float16)[None].contiguous()

    def extract(n):
        tokens, p, d = clip.vision_tokens(images[:n].to(dev))
        tf = clip.text_tokens(text[:n].to(dev), pooled_row=True)           # (n, 16, D)
        assert tf.shape == (n, 16, d)
        return ops.text_aware_pool(tokens, tf, pos_t, temp.to(dev), n, 1, 1, n_patches=p)[0], tf

    small, tf_small = extract(n_small)
    with torch.no_grad():
        x = R.clip_transform(images[:n_small])
        pf = clip_ref.patch_features(x, sd, "clip_model.", clip_ref.gelu_erf)                       # (4, 256, 768) normalised
        tf = clip_ref.text_features(text[:n_small], 15, sd, "clip_model.", clip_ref.gelu_erf)        # (4, 15, 768) normalised
        pooled = clip_ref.encode_text(text[:n_small], sd, "clip_model.", clip_ref.gelu_erf)
        pooled = pooled / pooled.norm(dim=-1, keepdim=True)
        tf16 = torch.cat([tf, pooled[:, None]], dim=1)
        ref = R.text_aware_visual_extraction(pf, tf16, temp[0], pos)
    got_tf = tf_small.cpu() / tf_small.cpu().norm(dim=-1, keepdim=True)
    assert _cos(got_tf, tf16) >= 0.9999
    assert _cos(got_tf[:, 15], tf16[:, 15]) >= 0.9999                      # the pooled row on its own
    err = (small.cpu() - ref).abs().max().item()
    print(f"OpenCLIP-variant extraction: cosine {_cos(small.cpu(), ref):.6f}, max abs err {err:.2e} (|ref| max {ref.abs().max():.2f})")
    # temperature 0.005 turns a 1e-4 error of a cosine into 2 % of a softmax weight (see the kernel-level test); measured
    # 4.8e-4 on this case, bound = four times that
    assert _cos(small.cpu(), ref) >= 0.9999 and err <= 2e-3
    big, _ = extract(n_big)
    # batch-size independent up to summation order: which rows of a GEMM are multiplied split-K depends on the row count
    torch.testing.assert_close(big[:n_small], small, rtol=0, atol=2e-3)
    assert torch.isfinite(big).all()


def test_long_history_rollout_64_episodes_window_32(dev):
    """Long-horizon configuration: 64 concurrent episodes per GPU, history window 32, cached K/V while the window fills
    and recompute once it slides.  Two of the episodes against single-episode rollouts of the CPU oracle."""
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.engine import BatchedRollout
    mc = ModelConfig(clip_model="tiny-test", transformer_dim=128, transformer_layers=2, transformer_heads=2, vision_pooling_output_dim=128,
                     text_pooling_output_dim=64, proprio_hidden_dim=64, proprio_output_dim=32, pooling_heads=4)
    sc = SharedConfig(image_size=112, seq_length=32, action_horizon=4)
    model, sd = _model(mc, sc, 21, dev)
    d = model.device_model()
    B, steps = 64, 36
    rng = np.random.default_rng(13)
    images = torch.from_numpy(rng.integers(0, 256, (steps, B, 2, 3, 112, 112), dtype=np.uint8))
    proprio = torch.from_numpy(rng.standard_normal((steps, B, 10)).astype(np.float32))
    text = GI._tokens(rng, B, (4, 7, 9))
    tf = d.clip.text_tokens_cached(text)
    batch = BatchedRollout(d, B)
    dims = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                  action_dim=mc.action_dim, action_horizon=sc.action_horizon, seq_length=sc.seq_length)
    oracles = {e: R.Rollout(sd, dims) for e in (0, 63)}
    worst = 0.0
    for t in range(steps):
        got = batch.step(images[t].reshape(B * 2, 3, 112, 112).to(dev), tf, proprio[t].to(dev)).cpu()
        for e, ro in oracles.items():
            ref = ro.step(R.clip_transform(images[t, e]), text[e], proprio[t, e])
            worst = max(worst, (got[e] - ref).abs().max().item())
    print(f"64 episodes x window 32: max |action - oracle| over {steps} steps = {worst:.2e}")
    assert worst <= 1e-3


def _write_synthetic_merges(path, n=49152 - 256 - 2):
    """A merges file with the real file's shape (header + 48894 pairs, so start/end-of-text get ids 49406 / 49407) but made-up
    pairs: the real vocabulary is not redistributed.  The first pairs are real English merges so prompts shorten a little."""
    from otter_amd.tokenizer import bytes_to_unicode
    base = list(bytes_to_unicode().values())
    syms = base + [b + "</w>" for b in base]
    pairs = [("t", "h"), ("th", "e</w>"), ("p", "i"), ("pi", "c"), ("pic", "k</w>"), ("u", "p</w>"), ("c", "u"), ("b", "e</w>"), ("cu", "be</w>")]
    seen = set(pairs)
    i = 0
    while len(pairs) < n:
        a, b = syms[100 + (i // 400) % 400], syms[101 + i % 400]      # 160,000 candidate pairs of mostly non-ASCII symbols
        i += 1
        if (a, b) not in seen:
            seen.add((a, b))
            pairs.append((a, b))
    with open(path, "w", encoding="utf-8") as fh:
        fh.write("#version: synthetic\n")
        fh.write("\n".join(f"{a} {b}" for a, b in pairs) + "\n")


def test_otter_inference_end_to_end(dev, tmp_path, monkeypatch):
    """The rollout interface as a robot script uses it: checkpoint folder on disk -> OtterInference -> PIL frames, prompt,
    pose and gripper in -> absolute action chunk out; six calls across the window boundary against the CPU oracle fed
    with PIL-resized frames; then reset()."""
    from PIL import Image
    from otter_amd import tokenizer as tk
    from otter_amd.action_utils import convert_abs_action
    from otter_amd.checkpoint import save_checkpoint
    from otter_amd.config import ExperimentConfig
    from otter_amd.otter_interface import OtterInference
    from otter_amd.vision_tf import crop2square, pil_to_uint8_chw
    from scipy.spatial.transform import Rotation
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, sd = _model(mc, sc, case["seed"], torch.device("cpu"))
    save_checkpoint(model, ExperimentConfig(model_cfg=mc, shared_cfg=sc), str(tmp_path), 3)
    bpe = tmp_path / "merges.txt"
    _write_synthetic_merges(str(bpe))
    monkeypatch.setenv("OTTER_CLIP_BPE", str(bpe))
    monkeypatch.setattr(tk, "_default", None)
    policy = OtterInference(str(tmp_path), 3, device="cuda")
    assert sorted(policy.model.camera_keys) == policy.model.camera_keys and policy.current_proprio is None
    prompt = "pick up the cube"
    ids = tk.tokenize([prompt])[0]
    assert ids[0] == 49406 and int(ids.max()) == 49407 and int((ids != 0).sum()) == 6        # SOT pick up the cube EOT
    rng = np.random.default_rng(3)
    dims = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                  action_dim=mc.action_dim, action_horizon=sc.action_horizon, seq_length=sc.seq_length)
    ro = R.Rollout(sd, dims)
    worst = 0.0
    for step in range(sc.seq_length + 2):
        frames = {k: Image.fromarray(rng.integers(0, 256, (180, 320, 3), dtype=np.uint8)) for k in policy.model.camera_keys}
        pose = rng.standard_normal(6) * 0.3
        grip = rng.uniform(-0.2, 1.2, 1)
        got = policy(frames, prompt, pose, grip)
        assert got.shape == (sc.action_horizon, mc.action_dim)
        rot = Rotation.from_euler("XYZ", pose[3:]).as_matrix()[:2].reshape(6)
        prop = np.concatenate([pose[:3], rot, np.clip(grip, 0, 1)])
        np.testing.assert_allclose(policy.current_proprio, prop, rtol=0, atol=1e-12)
        imgs = torch.stack([pil_to_uint8_chw(crop2square(frames[k]).resize((sc.image_size, sc.image_size), Image.Resampling.BICUBIC))
                            for k in policy.model.camera_keys])
        ref = ro.step(R.clip_transform(imgs), ids, torch.tensor(prop).float()).numpy()
        ref_abs = convert_abs_action(ref[None].astype(np.float64), prop[None, None])[0]
        worst = max(worst, float(np.abs(got - ref_abs).max()))
    print(f"OtterInference end to end: max |absolute action - oracle| = {worst:.2e}")
    assert worst <= 1e-3
    policy.reset()
    assert policy.current_proprio is None and policy.model.cache_size == 0
    with pytest.raises(AssertionError):
        policy({"image_primary": frames["image_primary"]}, prompt, pose, grip)              # a camera is missing


def test_otter_inference_sim_end_to_end(dev, tmp_path, monkeypatch):
    """The simulator interface (``otter_interface.py:188-307``): frames already at the model's resolution (``to_tensor`` +
    normalisation only), the 10-d proprioception passed through, delta chunks returned as predicted; against the CPU
    oracle across the window boundary, then ``reset()``."""
    from PIL import Image
    from otter_amd import tokenizer as tk
    from otter_amd.checkpoint import save_checkpoint
    from otter_amd.config import ExperimentConfig
    from otter_amd.otter_interface import OtterInferenceSim
    from otter_amd.vision_tf import pil_to_uint8_chw
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, sd = _model(mc, sc, case["seed"], torch.device("cpu"))
    save_checkpoint(model, ExperimentConfig(model_cfg=mc, shared_cfg=sc), str(tmp_path), 7)
    bpe = tmp_path / "merges.txt"
    _write_synthetic_merges(str(bpe))
    monkeypatch.setenv("OTTER_CLIP_BPE", str(bpe))
    monkeypatch.setattr(tk, "_default", None)
    policy = OtterInferenceSim(str(tmp_path), 7, device="cuda")
    prompt = "pick up the cube"
    ids = tk.tokenize([prompt])[0]
    rng = np.random.default_rng(5)
    dims = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                  action_dim=mc.action_dim, action_horizon=sc.action_horizon, seq_length=sc.seq_length)
    ro = R.Rollout(sd, dims)
    worst = 0.0
    for step in range(sc.seq_length + 2):
        frames = {k: Image.fromarray(rng.integers(0, 256, (sc.image_size, sc.image_size, 3), dtype=np.uint8)) for k in policy.model.camera_keys}
        prop = rng.standard_normal(mc.proprio_input_dim)
        got = policy(frames, prompt, prop)
        assert isinstance(got, np.ndarray) and got.shape == (sc.action_horizon, mc.action_dim)
        imgs = torch.stack([pil_to_uint8_chw(frames[k]) for k in policy.model.camera_keys])
        ref = ro.step(R.clip_transform(imgs), ids, torch.tensor(prop).float()).numpy()
        worst = max(worst, float(np.abs(got - ref).max()))
    print(f"OtterInferenceSim end to end: max |action - oracle| = {worst:.2e}")
    assert worst <= 1e-3
    policy.reset()
    assert policy.model.cache_size == 0
    with pytest.raises(AssertionError):
        policy({"image_primary": frames["image_primary"]}, prompt, prop)                    # a camera is missing


def test_small_batch_encoder_graph_matches_eager(dev, monkeypatch):
    """Steps of a few episodes replay a captured HIP graph of the encoder (launch-bound otherwise): same bits as the eager
    launches over a rollout across the window boundary, with fresh inputs every step and a new graph when the shapes change."""
    from otter_amd import engine
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, _ = _model(mc, sc, case["seed"], dev)
    dm = model.device_model()
    cams, res = len(model.camera_keys), sc.image_size
    g = torch.Generator().manual_seed(11)
    for n in (2, 1):
        ro_graph, ro_eager = engine.BatchedRollout(dm, n), engine.BatchedRollout(dm, n)
        for _ in range(sc.seq_length + 3):
            imgs = torch.randint(0, 256, (n * cams, 3, res, res), dtype=torch.uint8, generator=g).to(dev)
            tf = torch.randn(n, mc.first_k_tokens, dm.clip.embed, generator=g).to(dev)
            prop = torch.randn(n, mc.proprio_input_dim, generator=g).to(dev)
            monkeypatch.setattr(engine, "_USE_GRAPH", True)
            got = ro_graph.step(imgs, tf, prop).clone()
            monkeypatch.setattr(engine, "_USE_GRAPH", False)
            ref = ro_eager.step(imgs, tf, prop)
            assert torch.equal(got, ref)
        assert ro_graph._graph is not None and ro_eager._graph is None
        assert ro_graph._graph.key[0] == (n * cams, 3, res, res)


def test_bf16_towers_and_batch_independence(dev):
    """bfloat16 operands for the vision tower (the alternative to the fp16 default; the text tower keeps fp16 operands:
    its bf16 rounding alone costs 1.7e-3, profiles/r03_bf16_error_attribution.txt): visual tokens reach cosine 0.9999
    against the fp32 oracle and the actions of four observations stay within the path's 1e-3.  The all-bf16 setting
    (`clip_text_dtype = bfloat16`) is measured beside it and misses the tolerance -- the reason for the split.
    Also: an observation's result does not depend on the batch it is computed in."""
    from otter_amd.config import ModelConfig, SharedConfig
    mc, sc = ModelConfig(), SharedConfig()
    model, sd = _model(mc, sc, 0, dev)
    model.clip_compute_dtype = torch.bfloat16
    assert model.clip_device().text_dtype == torch.float16 and model.clip_device().dtype == torch.bfloat16
    rng = np.random.default_rng(8)
    images = torch.from_numpy(rng.integers(0, 256, (4, 1, 2, 3, 224, 224), dtype=np.uint8))
    text = GI._tokens(rng, 4, (5, 9))
    proprio = torch.from_numpy(rng.standard_normal((4, 1, 10)).astype(np.float32))
    pf, tf = model.extract_clip_features(images[:1].to(dev), text[:1])
    with torch.no_grad():
        ref_pf = clip_ref.patch_features(R.clip_transform(images[0, 0]), sd)
    assert min(_cos(pf[c][0].cpu(), ref_pf[c]) for c in range(2)) >= 0.9999
    acts = model(images.to(dev), text, proprio.to(dev)).cpu()
    with torch.no_grad():
        ref = R.forward(images, text, proprio, sd, R.Dims())
    err = (acts - ref).abs().max().item()
    alone = model(images[2:3].to(dev), text[2:3], proprio[2:3].to(dev)).cpu()
    torch.testing.assert_close(alone[0], acts[2], rtol=0, atol=5e-4)       # batch independent up to the split-K summation order
    model.clip_text_dtype = torch.bfloat16                                   # everything in bf16
    err_all = (model(images.to(dev), text, proprio.to(dev)).cpu() - ref).abs().max().item()
    print(f"bf16 vision tower + fp16 text tower: max |action - oracle| = {err:.2e}; text tower in bf16 too: {err_all:.2e}")
    assert err <= 1e-3
    assert err_all <= 4e-3            # not a supported configuration: recorded, not held to the tolerance
