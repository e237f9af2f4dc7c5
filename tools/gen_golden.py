"""Generate ``tests/golden/*.npz`` by running the REFERENCE's own modules on seeded inputs.

Run in the build container only (it needs ``/root/reference``):

    python tools/gen_golden.py

The fixtures hold inputs' seeds and the reference's outputs -- data, not source.  Inputs
are regenerated in the tests from the same seeds (``tests/golden_inputs.py``), weights from
``otter_amd.params.random_state_dict`` (loaded strictly into the reference modules here,
which also checks the checkpoint key layout).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))

import _ref_shims  # noqa: E402

_ref_shims.install()

from otter.policy import models as ref_models            # noqa: E402
from otter.policy import transformer as ref_tf           # noqa: E402
from otter.policy import vision_tf as ref_vtf            # noqa: E402
from otter.policy import otter as ref_otter              # noqa: E402
from otter.dataset import utils as ref_utils             # noqa: E402

from otter_amd.config import ModelConfig, SharedConfig   # noqa: E402
from otter_amd.params import (otter_specs, pooling_specs, policy_specs, action_head_specs,  # noqa: E402
                              proprio_specs, random_state_dict)
import golden_inputs as GI                                # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def npy(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ text-aware visual extraction
def gen_tave():
    out = {}
    for case in GI.TAVE_CASES:
        patches, text = GI.tave_inputs(case)
        mod = ref_models.TextAwareVisualExtraction(case["P"], case["D"], temperature=case["temperature"])
        with torch.no_grad():
            y = mod(patches, text)
        out[case["name"]] = npy(y)
        out[case["name"] + "__pos_emb_rows"] = npy(mod.pos_emb[[0, 1, case["P"] - 1]])
    save("tave", **out)


# ------------------------------------------------------------------ attention pooling
def gen_pooling():
    out = {}
    for case in GI.POOL_CASES:
        specs = pooling_specs("", case["D_in"], case["D_out"], case["layers"], case["readouts"])
        sd = random_state_dict(specs, seed=case["seed"])
        mod = ref_tf.AttentionPooling(case["D_in"], case["D_out"], case["heads"], case["layers"], case["readouts"])
        mod.load_state_dict(sd, strict=True)
        x, mask = GI.pool_inputs(case)
        with torch.no_grad():
            out[case["name"]] = npy(mod(x, mask))
    save("attention_pooling", **out)


# ------------------------------------------------------------------ policy transformer / heads / tables
def gen_policy():
    out = {}
    for case in GI.POLICY_CASES:
        mc = ModelConfig(transformer_dim=case["dim"], transformer_heads=case["heads"],
                         transformer_layers=case["layers"], transformer_expansion_factor=case["expansion"])
        sd = random_state_dict(policy_specs("", case["dim"], case["heads"], case["layers"], case["expansion"]),
                               seed=case["seed"])
        mod = ref_tf.CausalTransformer(mc)
        mod.load_state_dict(sd, strict=True)
        x = GI.policy_inputs(case)
        with torch.no_grad():
            y = mod(x)
            # prefix consistency: causal => first t outputs do not depend on later inputs
            y_prefix = mod(x[:, : case["T"] // 2])
        out[case["name"]] = npy(y)
        out[case["name"] + "__prefix"] = npy(y_prefix)
    save("causal_transformer", **out)

    out = {}
    pe = ref_tf.SinusoidalPositionalEmbedding(12, 768)
    out["sinusoidal_12x768"] = npy(pe.pe)
    out["sinusoidal_5x8"] = npy(ref_tf.SinusoidalPositionalEmbedding(5, 8).pe)
    out["sincos_7x8"] = npy(ref_models.sincos_position_embedding(7, 8))
    out["sincos_256x768_rows_0_1_100_255"] = npy(ref_models.sincos_position_embedding(256, 768)[[0, 1, 100, 255]])
    save("position_tables", **out)

    out = {}
    for case in GI.HEAD_CASES:
        sd = random_state_dict(action_head_specs("", case["dim"], case["A"], case["H"]), seed=case["seed"])
        mod = ref_models.ActionHead(case["dim"], case["A"], case["H"])
        mod.load_state_dict(sd, strict=True)
        x = GI.head_inputs(case)
        with torch.no_grad():
            out[case["name"]] = npy(mod(x))
    for case in GI.PROPRIO_CASES:
        sd = random_state_dict(proprio_specs("", case["inp"], case["hidden"], case["out"]), seed=case["seed"])
        mod = ref_models.ProprioceptionEncoder(case["inp"], case["hidden"], case["out"])
        mod.load_state_dict(sd, strict=True)
        x = GI.proprio_inputs(case)
        with torch.no_grad():
            out[case["name"]] = npy(mod(x))
    save("heads", **out)


# ------------------------------------------------------------------ host-side helpers
def gen_host():
    out = {}
    toks = GI.text_tokens()
    out["mask_2d"] = npy(ref_otter.create_text_mask(toks, 49406, 49407, 15))
    out["mask_1d"] = npy(ref_otter.create_text_mask(toks[1], 49406, 49407, 15))
    out["mask_k5"] = npy(ref_otter.create_text_mask(toks, 49406, 49407, 5))
    u8, f_big, f_unit = GI.transform_inputs()
    out["clip_transform_u8"] = npy(ref_vtf.clip_transform(u8))
    out["clip_transform_float_gt1"] = npy(ref_vtf.clip_transform(f_big))
    out["clip_transform_float_unit"] = npy(ref_vtf.clip_transform(f_unit))
    for i, (act, prop) in enumerate(GI.abs_action_inputs()):
        out[f"abs_action_{i}"] = ref_utils.convert_abs_action(act.copy(), prop.copy())
    a10 = GI.action10_inputs()
    out["action_10d_to_7d"] = np.stack([ref_utils.action_10d_to_7d(a.copy()) for a in a10])
    from PIL import Image
    sizes = []
    for (w, h) in GI.CROP_SIZES:
        im = ref_vtf.crop2square(Image.new("RGB", (w, h)))
        sizes.append(im.size)
    out["crop2square_sizes"] = np.asarray(sizes)
    save("host", **out)


# ------------------------------------------------------------------ end-to-end OTTER with a tiny CLIP
def gen_e2e():
    out = {}
    for case in GI.E2E_CASES:
        mc, sc = GI.e2e_configs(case)
        sd = random_state_dict(otter_specs(mc, sc), seed=case["seed"])
        model = ref_otter.OTTER(mc, sc)
        model.load_state_dict(sd, strict=True)          # strict: pins the key layout and shapes
        model.eval()
        images, text, proprio, gt = GI.e2e_inputs(case)
        # batched forward (actions) + loss + gradient norms (mirrors tests/test_dummy_model.py:46-66)
        actions = model(images, text, proprio)
        loss = model(images, text, proprio, gt)
        loss.backward()
        out[case["name"] + "__actions"] = npy(actions)
        out[case["name"] + "__loss"] = npy(loss)
        out[case["name"] + "__gradnorms"] = np.asarray([
            model.visual_extraction[0].temperature.grad.norm().item(),
            model.vision_poolings[0].query.grad.norm().item(),
            model.text_pooling.query.grad.norm().item(),
            list(model.action_head.parameters())[0].grad.norm().item()])
        # intermediate: transformer input (forward_encoder), for localising a mismatch
        with torch.no_grad():
            fimg = ref_vtf.clip_transform(images)
            mask = ref_otter.create_text_mask(text, model.sot_token, model.eot_token, model.first_k_tokens) \
                if model.pool_true_text else None
            out[case["name"] + "__encoder"] = npy(model.forward_encoder(fimg, text, proprio, mask))
            pf, tf = model.extract_clip_features(fimg, text)
            out[case["name"] + "__patch_feat_cam0_first"] = npy(pf[0][0, :4])
            out[case["name"] + "__text_feat_first"] = npy(tf[0])
        # rollout across the window boundary (otter.py:293-337)
        model.reset_cache()
        steps = []
        r_images, r_text, r_proprio = GI.rollout_inputs(case)
        for t in range(r_images.shape[0]):
            m1 = ref_otter.create_text_mask(r_text, model.sot_token, model.eot_token, model.first_k_tokens) \
                if model.pool_true_text else None
            a = model.forward_inference(ref_vtf.clip_transform(r_images[t]), r_text, r_proprio[t], m1)
            steps.append(npy(a))
            assert model.cache_size == min(t + 1, sc.seq_length)
        out[case["name"] + "__rollout"] = np.stack(steps)
        model.reset_cache()
        a0 = model.forward_inference(ref_vtf.clip_transform(r_images[0]), r_text, r_proprio[0],
                                     ref_otter.create_text_mask(r_text, model.sot_token, model.eot_token, model.first_k_tokens)
                                     if model.pool_true_text else None)
        assert np.array_equal(npy(a0), steps[0]), "reset_cache must restore the first-step result"
        out[case["name"] + "__sd_probe"] = GI.sd_probe(sd)
    save("otter_e2e", **out)


if __name__ == "__main__":
    gen_tave()
    gen_pooling()
    gen_policy()
    gen_host()
    gen_e2e()
