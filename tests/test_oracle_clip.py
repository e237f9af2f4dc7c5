"""CLIP towers of the oracle vs an independent implementation (``transformers.CLIPModel``).

The OpenAI ``clip`` package is not part of the reference tree, and the reference's tests hold no vectors for
it, so the towers are checked against the Hugging Face implementation built offline from a local config,
with its weights mapped into the OpenAI key layout the checkpoints use.
"""
import numpy as np
import pytest
import torch

from oracle import clip_ref

transformers = pytest.importorskip("transformers")


def _hf_to_openai(hf_sd, n_vis, n_txt):
    sd = {}
    p = "clip_model."
    sd[p + "visual.class_embedding"] = hf_sd["vision_model.embeddings.class_embedding"]
    sd[p + "visual.conv1.weight"] = hf_sd["vision_model.embeddings.patch_embedding.weight"]
    sd[p + "visual.positional_embedding"] = hf_sd["vision_model.embeddings.position_embedding.weight"]
    sd[p + "visual.ln_pre.weight"] = hf_sd["vision_model.pre_layrnorm.weight"]
    sd[p + "visual.ln_pre.bias"] = hf_sd["vision_model.pre_layrnorm.bias"]
    sd[p + "visual.ln_post.weight"] = hf_sd["vision_model.post_layernorm.weight"]
    sd[p + "visual.ln_post.bias"] = hf_sd["vision_model.post_layernorm.bias"]
    sd[p + "visual.proj"] = hf_sd["visual_projection.weight"].t().contiguous()
    sd[p + "text_projection"] = hf_sd["text_projection.weight"].t().contiguous()
    sd[p + "token_embedding.weight"] = hf_sd["text_model.embeddings.token_embedding.weight"]
    sd[p + "positional_embedding"] = hf_sd["text_model.embeddings.position_embedding.weight"]
    sd[p + "ln_final.weight"] = hf_sd["text_model.final_layer_norm.weight"]
    sd[p + "ln_final.bias"] = hf_sd["text_model.final_layer_norm.bias"]
    for hf_root, oa_root, n in (("vision_model", p + "visual.transformer", n_vis), ("text_model", p + "transformer", n_txt)):
        for i in range(n):
            h = f"{hf_root}.encoder.layers.{i}."
            o = f"{oa_root}.resblocks.{i}."
            sd[o + "attn.in_proj_weight"] = torch.cat([hf_sd[h + f"self_attn.{x}_proj.weight"] for x in "qkv"], 0)
            sd[o + "attn.in_proj_bias"] = torch.cat([hf_sd[h + f"self_attn.{x}_proj.bias"] for x in "qkv"], 0)
            sd[o + "attn.out_proj.weight"] = hf_sd[h + "self_attn.out_proj.weight"]
            sd[o + "attn.out_proj.bias"] = hf_sd[h + "self_attn.out_proj.bias"]
            for a, b in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
                sd[o + a + ".weight"] = hf_sd[h + b + ".weight"]
                sd[o + a + ".bias"] = hf_sd[h + b + ".bias"]
    return sd


# toy width, and the real ViT-L/14 geometry (width 1024 / 16 heads / 257 tokens; text width 768 / 12 heads) at 2 layers
GEOMETRIES = {
    "toy": dict(tw=128, ti=512, tl=2, th=2, vw=128, vi=512, vl=3, vh=2, res=56, proj=64, n_img=3),
    "vit_l14_geometry": dict(tw=768, ti=3072, tl=2, th=12, vw=1024, vi=4096, vl=2, vh=16, res=224, proj=768, n_img=2),
}


@pytest.mark.parametrize("geometry", list(GEOMETRIES))
@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_clip_towers_match_transformers(act, geometry):
    from transformers import CLIPConfig, CLIPModel
    g = GEOMETRIES[geometry]
    if geometry != "toy" and act == "gelu":
        pytest.skip("the activation is covered at the toy width")
    torch.manual_seed(0)
    cfg = CLIPConfig(
        text_config=dict(hidden_size=g["tw"], intermediate_size=g["ti"], num_hidden_layers=g["tl"], num_attention_heads=g["th"],
                         vocab_size=49408, max_position_embeddings=77, hidden_act=act, eos_token_id=49407,
                         bos_token_id=49406, pad_token_id=0),
        vision_config=dict(hidden_size=g["vw"], intermediate_size=g["vi"], num_hidden_layers=g["vl"], num_attention_heads=g["vh"],
                           image_size=g["res"], patch_size=14, hidden_act=act),
        projection_dim=g["proj"])
    hf = CLIPModel(cfg).eval()
    # give biases / LN parameters non-trivial values
    with torch.no_grad():
        for p_ in hf.parameters():
            if p_.dim() == 1:
                p_.add_(torch.randn_like(p_) * 0.1)
    sd = _hf_to_openai(hf.state_dict(), g["vl"], g["tl"])
    f = clip_ref.quick_gelu if act == "quick_gelu" else clip_ref.gelu_erf
    images = torch.randn(g["n_img"], 3, g["res"], g["res"])
    tokens = torch.zeros(2, 77, dtype=torch.long)
    tokens[0, :5] = torch.tensor([49406, 5, 77, 901, 49407])
    tokens[1, :9] = torch.tensor([49406, 15, 7, 91, 3, 1000, 40000, 2, 49407])

    captured = {}
    hook = hf.vision_model.encoder.layers[-1].self_attn.register_forward_hook(
        lambda m, i, o: captured.__setitem__("attn", o[0]))
    with torch.no_grad():
        img_hf = hf.get_image_features(pixel_values=images)
        txt_hf = hf.get_text_features(input_ids=tokens)
        if not torch.is_tensor(img_hf):
            img_hf = img_hf.pooler_output
        if not torch.is_tensor(txt_hf):
            txt_hf = txt_hf.pooler_output
        img = clip_ref.encode_image(images, sd, act=f)
        txt = clip_ref.encode_text(tokens, sd, act=f)
        last = clip_ref.vision_last_attn(images, sd, act=f)
    hook.remove()
    assert last.shape == (g["n_img"], (g["res"] // 14) ** 2 + 1, g["vw"])
    np.testing.assert_allclose(img.numpy(), img_hf.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(txt.numpy(), txt_hf.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(last.numpy(), captured["attn"].numpy(), rtol=1e-4, atol=1e-5)


def test_text_positions_before_k_are_causal():
    """Only the first k positions of the text tower are read (otter.py:193); with the causal mask they do
    not depend on later tokens, which is what lets the device path compute k positions instead of 77."""
    from otter_amd.config import clip_arch
    from otter_amd.params import clip_specs, random_state_dict
    sd = random_state_dict(clip_specs(clip_arch("tiny-test")), seed=3)
    t = torch.zeros(1, 77, dtype=torch.long)
    t[0, :20] = torch.randint(1, 49000, (20,))
    t2 = t.clone()
    t2[0, 15:40] = torch.randint(1, 49000, (25,))
    a = clip_ref.text_transformer_out(t, sd, heads=1)[:, :15]
    b = clip_ref.text_transformer_out(t2, sd, heads=1)[:, :15]
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-6)
    c = clip_ref.text_transformer_out(t[:, :15], sd, heads=1)
    np.testing.assert_allclose(a.numpy(), c.numpy(), rtol=1e-5, atol=1e-6)
