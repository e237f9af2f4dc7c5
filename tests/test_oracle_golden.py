"""The CPU oracle against fixtures produced by the reference's own modules (tools/gen_golden.py)."""
import numpy as np
import pytest
import torch

import golden_inputs as GI
from oracle import otter_ref as R
from otter_amd.params import (action_head_specs, otter_specs, policy_specs, pooling_specs, proprio_specs,
                              random_state_dict)

TOL = dict(rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("case", GI.TAVE_CASES, ids=lambda c: c["name"])
def test_tave(golden, case):
    g = golden("tave")
    p, t = GI.tave_inputs(case)
    pos = R.sincos_position_embedding(case["P"], case["D"])
    np.testing.assert_allclose(pos[[0, 1, case["P"] - 1]].numpy(), g[case["name"] + "__pos_emb_rows"], **TOL)
    y = R.text_aware_visual_extraction(p, t, torch.tensor(case["temperature"]), pos)
    np.testing.assert_allclose(y.numpy(), g[case["name"]], **TOL)


@pytest.mark.parametrize("case", GI.POOL_CASES, ids=lambda c: c["name"])
def test_attention_pooling(golden, case):
    g = golden("attention_pooling")
    sd = random_state_dict(pooling_specs("", case["D_in"], case["D_out"], case["layers"], case["readouts"]), seed=case["seed"])
    x, mask = GI.pool_inputs(case)
    y = R.attention_pooling(x, sd, "", case["heads"], mask)
    np.testing.assert_allclose(y.numpy(), g[case["name"]], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", GI.POLICY_CASES, ids=lambda c: c["name"])
def test_causal_transformer(golden, case):
    g = golden("causal_transformer")
    sd = random_state_dict(policy_specs("", case["dim"], case["heads"], case["layers"], case["expansion"]), seed=case["seed"])
    x = GI.policy_inputs(case)
    y = R.causal_transformer(x, sd, case["heads"], prefix="")
    np.testing.assert_allclose(y.numpy(), g[case["name"]], rtol=1e-4, atol=2e-5)
    half = case["T"] // 2
    yp = R.causal_transformer(x[:, :half], sd, case["heads"], prefix="")
    np.testing.assert_allclose(yp.numpy(), g[case["name"] + "__prefix"], rtol=1e-4, atol=2e-5)
    # causality: a prefix of the input gives the prefix of the output
    np.testing.assert_allclose(yp.numpy(), y[:, :half].numpy(), rtol=1e-4, atol=2e-5)


def test_position_tables(golden):
    g = golden("position_tables")
    np.testing.assert_allclose(R.sinusoidal_pe_table(12, 768).numpy(), g["sinusoidal_12x768"], **TOL)
    np.testing.assert_allclose(R.sinusoidal_pe_table(5, 8).numpy(), g["sinusoidal_5x8"], **TOL)
    np.testing.assert_allclose(R.sincos_position_embedding(7, 8).numpy(), g["sincos_7x8"], **TOL)
    np.testing.assert_allclose(R.sincos_position_embedding(256, 768)[[0, 1, 100, 255]].numpy(),
                               g["sincos_256x768_rows_0_1_100_255"], **TOL)


def test_heads(golden):
    g = golden("heads")
    for case in GI.HEAD_CASES:
        sd = random_state_dict(action_head_specs("", case["dim"], case["A"], case["H"]), seed=case["seed"])
        y = R.action_head(GI.head_inputs(case), sd, prefix="").reshape(case["B"], case["H"], -1)
        np.testing.assert_allclose(y.numpy(), g[case["name"]], rtol=1e-4, atol=1e-5)
    for case in GI.PROPRIO_CASES:
        sd = random_state_dict(proprio_specs("", case["inp"], case["hidden"], case["out"]), seed=case["seed"])
        y = R.proprio_encoder(GI.proprio_inputs(case), sd, prefix="")
        np.testing.assert_allclose(y.numpy(), g[case["name"]], rtol=1e-4, atol=1e-5)


def test_text_mask_and_transform(golden):
    g = golden("host")
    toks = GI.text_tokens()
    assert np.array_equal(R.create_text_mask(toks, 49406, 49407, 15).numpy(), g["mask_2d"])
    assert np.array_equal(R.create_text_mask(toks[1], 49406, 49407, 15).numpy(), g["mask_1d"])
    assert np.array_equal(R.create_text_mask(toks, 49406, 49407, 5).numpy(), g["mask_k5"])
    u8, f_big, f_unit = GI.transform_inputs()
    np.testing.assert_allclose(R.clip_transform(u8).numpy(), g["clip_transform_u8"], **TOL)
    np.testing.assert_allclose(R.clip_transform(f_big).numpy(), g["clip_transform_float_gt1"], **TOL)
    np.testing.assert_allclose(R.clip_transform(f_unit).numpy(), g["clip_transform_float_unit"], **TOL)


@pytest.mark.parametrize("case", GI.E2E_CASES, ids=lambda c: c["name"])
def test_end_to_end_tiny(golden, case):
    """Whole path (tiny CLIP) vs the reference's OTTER class: batched forward, loss, encoder output,
    and a rollout that crosses the history-window boundary."""
    g = golden("otter_e2e")
    n = case["name"]
    mc, sc = GI.e2e_configs(case)
    sd = random_state_dict(otter_specs(mc, sc), seed=case["seed"])
    np.testing.assert_allclose(GI.sd_probe(sd), g[n + "__sd_probe"], rtol=1e-12)
    dims = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads,
                  transformer_heads=mc.transformer_heads, action_dim=mc.action_dim,
                  action_horizon=sc.action_horizon, seq_length=sc.seq_length,
                  pool_true_text=mc.pool_true_text, use_mse=sc.mse_loss)
    images, text, proprio, gt = GI.e2e_inputs(case)
    with torch.no_grad():
        actions = R.forward(images, text, proprio, sd, dims)
        loss = R.forward(images, text, proprio, sd, dims, gt)
        fimg = R.clip_transform(images)
        mask = R.create_text_mask(text, dims.sot_token, dims.eot_token, dims.first_k_tokens) if dims.pool_true_text else None
        enc = R.forward_encoder(fimg, text, proprio, sd, dims, mask)
        pf, tf = R.extract_clip_features(fimg, text, sd, dims)
    np.testing.assert_allclose(pf[0][0, :4].numpy(), g[n + "__patch_feat_cam0_first"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tf[0].numpy(), g[n + "__text_feat_first"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(enc.numpy(), g[n + "__encoder"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(actions.numpy(), g[n + "__actions"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(loss.numpy(), g[n + "__loss"], rtol=1e-5)

    r_images, r_text, r_proprio = GI.rollout_inputs(case)
    ro = R.Rollout(sd, dims)
    m1 = R.create_text_mask(r_text, dims.sot_token, dims.eot_token, dims.first_k_tokens) if dims.pool_true_text else None
    steps = []
    for t in range(r_images.shape[0]):
        steps.append(ro.step(R.clip_transform(r_images[t]), r_text, r_proprio[t], m1).numpy())
        assert ro.cache_size == min(t + 1, sc.seq_length)
    np.testing.assert_allclose(np.stack(steps), g[n + "__rollout"], rtol=1e-4, atol=2e-5)
    ro.reset()
    a0 = ro.step(R.clip_transform(r_images[0]), r_text, r_proprio[0], m1).numpy()
    np.testing.assert_allclose(a0, steps[0], rtol=0, atol=0)
