"""CPU tests: host-side logic of the product (tokenizer, config, pose/action helpers, checkpoint contract) and the
C-ABI library's exported symbols.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import golden_inputs as GI

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "otter_hip.h")).read()
    declared = set(re.findall(r"\b(otter_[a-z0-9_]+)\s*\(", header))
    declared.discard("otter_readout_block")
    assert len(declared) >= 15
    from otter_amd import _lib
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libotter_hip.so does not export {name}"
    assert set(_lib.exported_names()) == declared, (set(_lib.exported_names()) ^ declared)
    assert lib.otter_hip_abi_version() == _lib.ABI_VERSION
    m = re.search(r"#define OTTER_HIP_ABI_VERSION (\d+)", header)
    assert int(m.group(1)) == _lib.ABI_VERSION


def test_abi_argument_validation_without_gpu():
    """Argument checks run before any launch, so they can be exercised on a machine without a GPU."""
    from otter_amd import _lib
    lib = _lib.load()
    rc = lib.otter_gemm(0, None, 0, None, 0, None, None, 0, None, 0, 1, 1, 1, 0, None)
    assert rc != 0 and b"null operand" in lib.otter_last_error()
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.otter_gemm(0, p, 100, p, 128, None, None, 0, p, 128, 4, 128, 100, 0, None)    # K % 64 != 0
    assert rc == 2 and b"multiple of 64" in lib.otter_last_error()
    rc = lib.otter_gemm(7, p, 64, p, 128, None, None, 0, p, 128, 4, 128, 64, 0, None)      # unknown dtype
    assert rc == 1
    rc = lib.otter_attn_fwd_d64(0, p, 192, p, 64, 1, 600, 1, 0, None)                      # L too long
    assert rc == 2 and b"592" in lib.otter_last_error()


def test_ops_refuse_cpu_tensors():
    from otter_amd import ops
    with pytest.raises(ops.OtterHipError):
        ops.layernorm(torch.zeros(2, 64), None, None)
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.otter import OTTER
    m = OTTER(ModelConfig(clip_model="tiny-test"), SharedConfig(image_size=112))
    with pytest.raises(ops.OtterHipError):
        m.forward_inference(torch.zeros(2, 3, 112, 112), torch.zeros(77, dtype=torch.int32), torch.zeros(10))


# ---------------------------------------------------------------- checkpoint contract
def test_state_dict_layout_and_counts():
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.params import otter_specs
    specs = otter_specs(ModelConfig(), SharedConfig())
    numel = lambda s: int(np.prod(s.shape)) if s.shape else 1     # noqa: E731
    clip = sum(numel(s) for s in specs if s.key.startswith("clip_model."))
    trainable = sum(numel(s) for s in specs if s.kind == "param")
    policy = sum(numel(s) for s in specs if s.key.startswith("policy."))
    assert clip == 427_616_513            # OpenAI ViT-L/14
    assert trainable == 31_055_290        # gradient all-reduce message of the training step (x4 bytes = 124.2 MB)
    assert policy == 28_372_992
    keys = [s.key for s in specs]
    assert len(keys) == len(set(keys))
    for k in ("clip_model.visual.transformer.resblocks.23.attn.in_proj_weight", "text_pooling.blocks.1.attention.k_proj_weight",
              "visual_extraction.1.pos_emb", "vision_poolings.0.query", "proprio_encoder.mlp.4.bias", "input_projection.weight",
              "pe.pe", "policy.layers.7.attention.value.bias", "action_head.mlp.2.weight"):
        assert k in keys


def test_model_object_mirrors_reference_attributes(tmp_path):
    from otter_amd.config import ExperimentConfig, ModelConfig, SharedConfig, dump_run_yaml, load_run_yaml
    from otter_amd.otter import OTTER
    from otter_amd.otter_interface import load_state_dict_flexible
    mc, sc = GI.e2e_configs(GI.E2E_CASES[0])
    m = OTTER(mc, sc)
    for attr in ("camera_keys", "action_horizon", "action_dim", "seq_length", "pool_true_text", "sot_token", "eot_token",
                 "first_k_tokens", "proprio_input_dim", "cached_transformer_input", "cache_size", "image_size"):
        assert hasattr(m, attr)
    assert m.camera_keys == ["image_primary", "image_wrist"] and m.sot_token == 49406 and m.eot_token == 49407
    assert m.cached_transformer_input is None and m.cache_size == 0
    assert all(not p.requires_grad for n, p in m.named_parameters() if n.startswith("clip_model."))
    assert m.visual_extraction[0].temperature.requires_grad and float(m.visual_extraction[0].temperature.detach()) == pytest.approx(0.07)
    assert "visual_extraction.0.pos_emb" in dict(m.named_buffers())
    # DDP-style checkpoints load, wrong keys are rejected (strict)
    sd = {"module." + k: v for k, v in m.state_dict().items()}
    load_state_dict_flexible(OTTER(mc, sc), sd)
    bad = dict(m.state_dict())
    bad.pop("pe.pe")
    with pytest.raises(RuntimeError):
        OTTER(mc, sc).load_state_dict(bad)
    # run.yaml round trip, including the reference's python-object tags and an unknown data-layer reference
    cfg = ExperimentConfig(model_cfg=mc, shared_cfg=sc)
    path = tmp_path / "run.yaml"
    dump_run_yaml(cfg, path)
    text = path.read_text()
    assert "!!python/object:otter.util.args.ModelConfig" in text
    path.write_text(text + "dataset_cfg: !!python/object:otter.util.args.DatasetConfig\n"
                           "  restructure: !!python/name:otter.data.dataset.bridge_restructure ''\n"
                           "  sample_weights: !!python/tuple [1.5, 2.5]\ntrain: true\n")
    back = load_run_yaml(path)
    assert back.model_cfg == mc and back.shared_cfg == sc
    assert back.extras["train"] is True and back.extras["dataset_cfg"]["sample_weights"] == (1.5, 2.5)


# ---------------------------------------------------------------- host helpers vs the reference's outputs
def test_host_helpers_match_reference_fixtures(golden):
    from PIL import Image
    from otter_amd.action_utils import action_10d_to_7d, convert_abs_action
    from otter_amd.otter import create_text_mask
    from otter_amd.vision_tf import clip_transform_host, crop2square
    g = golden("host")
    toks = GI.text_tokens()
    assert np.array_equal(create_text_mask(toks, 49406, 49407, 15).numpy(), g["mask_2d"])
    assert np.array_equal(create_text_mask(toks[1], 49406, 49407, 15).numpy(), g["mask_1d"])
    assert np.array_equal(create_text_mask(toks, 49406, 49407, 5).numpy(), g["mask_k5"])
    u8, f_big, f_unit = GI.transform_inputs()
    np.testing.assert_allclose(clip_transform_host(u8).numpy(), g["clip_transform_u8"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(clip_transform_host(f_big).numpy(), g["clip_transform_float_gt1"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(clip_transform_host(f_unit).numpy(), g["clip_transform_float_unit"], rtol=1e-6, atol=1e-6)
    for i, (act, prop) in enumerate(GI.abs_action_inputs()):
        np.testing.assert_allclose(convert_abs_action(act.copy(), prop.copy()), g[f"abs_action_{i}"], rtol=1e-12, atol=1e-12)
    a10 = GI.action10_inputs()
    np.testing.assert_allclose(np.stack([action_10d_to_7d(a.copy()) for a in a10]), g["action_10d_to_7d"], rtol=1e-12, atol=1e-12)
    sizes = [crop2square(Image.new("RGB", wh)).size for wh in GI.CROP_SIZES]
    assert np.array_equal(np.asarray(sizes), g["crop2square_sizes"])


def test_proprio_preprocessing_and_action_decode():
    """otter_interface.py:60-84 and :113-133 on the host (no model needed)."""
    from scipy.spatial.transform import Rotation
    from otter_amd.otter_interface import OtterInference
    obj = OtterInference.__new__(OtterInference)
    pose = np.array([0.3, -0.2, 0.5, 0.1, -0.4, 1.2])
    t = obj._proprocess_proprio(pose, np.array([1.7]))
    R = Rotation.from_euler("XYZ", pose[3:]).as_matrix()
    expect = np.concatenate([pose[:3], R[0], R[1], [1.0]])
    np.testing.assert_allclose(obj.current_proprio, expect, rtol=0, atol=1e-15)
    assert t.dtype == torch.float32 and t.shape == (10,)
    # a zero-translation, identity-rotation delta chunk decodes to the current pose
    chunk = np.tile(np.concatenate([np.zeros(3), [1, 0, 0, 0, 1, 0], [0.5]]), (12, 1))
    out = obj._decode_actions(torch.from_numpy(chunk))
    np.testing.assert_allclose(out[:, :9], np.tile(expect[:9], (12, 1)), atol=1e-12)
    np.testing.assert_allclose(out[:, 9], 0.5)


# ---------------------------------------------------------------- tokenizer
def _toy_tokenizer():
    from otter_amd.tokenizer import SimpleTokenizer
    merges = [("t", "h"), ("th", "e</w>"), ("c", "u"), ("b", "e</w>"), ("cu", "be</w>"), ("p", "i"), ("pi", "c"), ("pic", "k</w>"),
              ("u", "p</w>"), ("r", "e"), ("re", "d</w>")]
    return SimpleTokenizer(merges=merges)


def test_bpe_on_synthetic_merges():
    from otter_amd.tokenizer import tokenize
    tk = _toy_tokenizer()
    sot, eot = tk.encoder["<|startoftext|>"], tk.encoder["<|endoftext|>"]
    assert (sot, eot) == (512 + 11, 512 + 12)        # vocabulary order: 256 bytes, 256 byte</w>, merges, specials
    ids = tk.encode("Pick  up the\nRED cube!")
    words = [tk.decoder[i] for i in ids]
    assert words == ["pick</w>", "up</w>", "the</w>", "red</w>", "cube</w>", "!</w>"]
    assert tk.decode(ids) == "pick up the red cube ! "
    # single characters, digits split one by one, apostrophe suffixes, html entities, unicode bytes
    assert [tk.decoder[i] for i in tk.encode("a 42 it's &amp;")] == ["a</w>", "4</w>", "2</w>", "i", "t</w>", "'", "s</w>", "&</w>"]   # "'s" is one regex piece; no merge joins it here
    assert tk.decode(tk.encode("café")) == "café "
    out = tokenize(["pick up the cube", ""], context_length=10, tokenizer=tk)
    assert out.dtype == torch.int32 and out.shape == (2, 10)
    assert out[0].tolist() == [sot] + tk.encode("pick up the cube") + [eot] + [0] * 4
    assert out[1].tolist() == [sot, eot] + [0] * 8
    with pytest.raises(RuntimeError):
        tokenize(["the " * 20], context_length=10, tokenizer=tk)
    cut = tokenize(["the " * 20], context_length=10, truncate=True, tokenizer=tk)
    assert cut[0, 0] == sot and cut[0, -1] == eot


def test_real_vocabulary_known_answer_if_available():
    """Known answer from the public CLIP documentation; runs only where the merges file is supplied."""
    from otter_amd.tokenizer import SimpleTokenizer, tokenize
    path = os.environ.get("OTTER_CLIP_BPE")
    if not path or not os.path.exists(path):
        pytest.skip("set OTTER_CLIP_BPE to bpe_simple_vocab_16e6.txt.gz to run (file is not redistributed)")
    tk = SimpleTokenizer(path)
    assert tokenize(["a photo of a cat"], tokenizer=tk)[0, :7].tolist() == [49406, 320, 1125, 539, 320, 2368, 49407]


def test_non_ascii_prompt_without_ftfy_is_reported(monkeypatch):
    """clip.tokenize runs ftfy.fix_text first; without ftfy a non-ASCII prompt may tokenise differently from the reference:
    warned about once per process, an error in strict mode; ASCII prompts are never affected."""
    import warnings
    from otter_amd import tokenizer
    try:
        import ftfy  # noqa: F401
        pytest.skip("ftfy is installed: prompts are repaired as the reference repairs them")
    except ImportError:
        pass
    tk = _toy_tokenizer()
    monkeypatch.setattr(tokenizer, "_warned_unrepaired", False)
    monkeypatch.delenv("OTTER_TOKENIZER_STRICT", raising=False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        tk.encode("pick up the cube")                                  # ASCII: silent
    with pytest.warns(tokenizer.UnrepairedTextWarning, match="ftfy"):
        tk.encode("the café table")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        tk.encode("naïve gripper")                                # once per process
    monkeypatch.setenv("OTTER_TOKENIZER_STRICT", "1")
    with pytest.raises(RuntimeError, match="ftfy"):
        tk.encode("the café table")
    tk.encode("pick up the cube")


def test_tokenizer_without_vocabulary_fails_loudly(monkeypatch):
    from otter_amd import tokenizer
    monkeypatch.delenv("OTTER_CLIP_BPE", raising=False)
    monkeypatch.setattr(tokenizer, "_default", None)
    with pytest.raises(FileNotFoundError):
        tokenizer.tokenize(["pick up the cube"])


# ------------------------------------------------------------------------------------------------ training host logic
def test_weight_decay_groups_follow_the_reference_rule():
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.otter import OTTER
    from otter_amd.trainer import add_weight_decay
    mc = ModelConfig(clip_model="tiny-test", transformer_dim=128, transformer_layers=2, transformer_heads=2, vision_pooling_output_dim=128,
                     text_pooling_output_dim=64, proprio_hidden_dim=64, proprio_output_dim=32, pooling_heads=4)
    model = OTTER(mc, SharedConfig(image_size=112, seq_length=4, action_horizon=4))
    no_decay, decay = add_weight_decay(model, 0.01)
    assert no_decay["weight_decay"] == 0.0 and decay["weight_decay"] == 0.01
    ids_nd, ids_d = {id(p) for p in no_decay["params"]}, {id(p) for p in decay["params"]}
    named = dict(model.named_parameters())
    trainable = {id(p) for p in named.values() if p.requires_grad}
    assert ids_nd | ids_d == trainable and not (ids_nd & ids_d)
    assert not any(n.startswith("clip_model.") and id(p) in trainable for n, p in named.items())
    assert id(named["policy.layers.0.pre_ln.weight"]) in ids_nd and id(named["input_projection.bias"]) in ids_nd
    assert id(named["text_pooling.blocks.0.attention.in_proj_bias"]) in ids_nd          # 1-D
    assert id(named["policy.layers.0.mlp.0.weight"]) in ids_d and id(named["text_pooling.query"]) in ids_d
    assert id(named["visual_extraction.0.temperature"]) in ids_d                          # 0-d: len(shape) != 1, so it decays (as upstream)


def test_cosine_schedule_matches_the_transformers_implementation():
    from transformers.optimization import get_cosine_schedule_with_warmup
    from otter_amd.trainer import cosine_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sched = get_cosine_schedule_with_warmup(opt, num_warmup_steps=5, num_training_steps=40)
    for step in range(45):
        assert abs(opt.param_groups[0]["lr"] - cosine_with_warmup(step, 5, 40)) < 1e-12
        opt.step()
        sched.step()


def test_loss_scale_state_machine():
    from otter_amd.trainer import LossScale
    s = LossScale(growth_interval=3)
    assert s.scale == 65536.0
    s.update(True)
    assert s.scale == 32768.0
    for _ in range(3):
        s.update(False)
    assert s.scale == 65536.0
    s.update(False); s.update(True); s.update(False); s.update(False)
    assert s.scale == 32768.0                                    # the clean-step counter restarted after the overflow
    t = LossScale()
    t.load_state_dict(s.state_dict())
    assert t.scale == s.scale
    assert LossScale(enabled=False).scale == 1.0


def test_flatten_parameters_aliases_the_module():
    from otter_amd.trainer import add_weight_decay, bucket_bounds, flatten_parameters
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.LayerNorm(3))
    before = {n: p.detach().clone() for n, p in lin.named_parameters()}
    groups = add_weight_decay(lin, 0.1)
    flat, spans, where = flatten_parameters(groups)
    assert flat.numel() == 4 * 64 and spans == [(0, 192, 0.0), (192, 256, 0.1)]       # every tensor starts on a 64-element boundary
    for n, p in lin.named_parameters():
        assert torch.equal(p.detach(), before[n])
        lo, hi = where[id(p)]
        assert p.data_ptr() == flat[lo:hi].data_ptr()
    flat.zero_()
    assert all(float(p.abs().sum()) == 0 for p in lin.parameters())
    assert bucket_bounds(10, 4) == [(0, 4), (4, 8), (8, 10)]


# ------------------------------------------------------------------------------------- rollout loop and checkpoints
def test_temporal_ensemble_and_receding_horizon_queue():
    from otter_amd.rollout_server import ActionQueue, to_robot_action
    a_dim = 10
    chunks = [np.full((12, a_dim), 100.0 * s) + np.arange(12)[:, None] for s in range(5)]     # chunk s predicts value 100 s + t
    q = ActionQueue(3, receding_horizon_control=False, k=0.05)
    got = [q.next_action(c)[0] for c in chunks]
    # step 0: only chunk 0 (t=0) -> 0.  step 1: chunk 0 (t=1) and chunk 1 (t=0), weights e^0, e^0.05.
    w2 = np.exp(0.05 * np.arange(2)); w2 /= w2.sum()
    w3 = np.exp(0.05 * np.arange(3)); w3 /= w3.sum()
    assert got[0] == 0.0
    assert abs(got[1] - (w2[0] * 1 + w2[1] * 100)) < 1e-12
    assert abs(got[2] - (w3[0] * 2 + w3[1] * 101 + w3[2] * 200)) < 1e-12
    # step 3: chunk 0 has been dropped (queue length 3); chunks 1, 2, 3 contribute their elements 2, 1, 0
    assert abs(got[3] - (w3[0] * 102 + w3[1] * 201 + w3[2] * 300)) < 1e-12
    r = ActionQueue(3, receding_horizon_control=True)
    seq = []
    for c in chunks:
        assert r.needs_prediction() == (len(seq) % 3 == 0)
        seq.append(r.next_action(c)[0])
    assert seq == [0.0, 1.0, 2.0, 300.0, 301.0]                  # executes 3 actions of a chunk, then takes the newest chunk
    act = np.array([0.1, 0.2, 0.3, 1, 0, 0, 0, 1, 0, 0.7])
    out = to_robot_action(act)
    assert out.shape == (7,) and np.allclose(out[:6], [0.1, 0.2, 0.3, 0, 0, 0]) and out[6] == 1.0
    assert to_robot_action(np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0.2]))[6] == 0.0
    with pytest.raises(ValueError):
        ActionQueue(0)


def test_checkpoint_round_trip_in_reference_layout(tmp_path):
    from otter_amd.checkpoint import load_checkpoint_state, save_checkpoint
    from otter_amd.config import ExperimentConfig, ModelConfig, SharedConfig, load_run_yaml
    from otter_amd.otter import OTTER
    from otter_amd.otter_interface import load_state_dict_flexible
    mc = ModelConfig(clip_model="tiny-test", transformer_dim=128, transformer_layers=2, transformer_heads=2, vision_pooling_output_dim=128,
                     text_pooling_output_dim=64, proprio_hidden_dim=64, proprio_output_dim=32, pooling_heads=4)
    sc = SharedConfig(image_size=112, seq_length=4, action_horizon=4)
    args = ExperimentConfig(model_cfg=mc, shared_cfg=sc)
    model = OTTER(mc, sc)
    with torch.no_grad():
        model.input_projection.bias.add_(1.5)
    path = save_checkpoint(model, args, str(tmp_path), 7)
    assert os.path.basename(path) == "checkpoint_7.pt" and os.path.exists(tmp_path / "run.yaml")
    back = load_run_yaml(str(tmp_path / "run.yaml"))
    assert back.model_cfg.clip_model == "tiny-test" and back.shared_cfg.seq_length == 4
    fresh = OTTER(back.model_cfg, back.shared_cfg)
    state = load_checkpoint_state(path)
    load_state_dict_flexible(fresh, {"module." + k: v for k, v in state.items()})        # as saved from a DDP wrapper
    for (n1, p1), (n2, p2) in zip(model.state_dict().items(), fresh.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2)


def test_published_clip_weights_load_into_the_model(tmp_path):
    """A tensors-only state_dict file in the OpenAI key layout (fp16 tensors plus the archive's three scalar entries) fills
    clip_model.*; a wrong shape or a missing tensor is an error, and a file that is not plain tensors is refused."""
    from otter_amd.clip_weights import load_clip_weights
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.otter import OTTER
    mc = ModelConfig(clip_model="tiny-test", transformer_dim=128, transformer_layers=2, transformer_heads=2, vision_pooling_output_dim=128,
                     text_pooling_output_dim=64, proprio_hidden_dim=64, proprio_output_dim=32, pooling_heads=4)
    sc = SharedConfig(image_size=112, seq_length=4, action_horizon=4)
    donor, model = OTTER(mc, sc), OTTER(mc, sc)
    g = torch.Generator().manual_seed(3)
    published = {}
    for k, v in donor.state_dict().items():
        if k.startswith("clip_model."):
            published[k[len("clip_model."):]] = (torch.randn(v.shape, generator=g) * 0.05).half()
    published.update(input_resolution=torch.tensor(112), context_length=torch.tensor(77), vocab_size=torch.tensor(49408))
    path = str(tmp_path / "clip.pt")
    torch.save({"state_dict": published}, path)
    before = {k: v.clone() for k, v in model.state_dict().items() if not k.startswith("clip_model.")}
    unused = load_clip_weights(model, path)
    assert sorted(unused) == ["context_length", "input_resolution", "vocab_size"]
    sd = model.state_dict()
    for k, v in published.items():
        if k not in unused:
            assert torch.equal(sd["clip_model." + k], v.float()), k
    for k, v in before.items():
        assert torch.equal(sd[k], v), k                                   # nothing outside clip_model moved
    bad = dict(published)
    bad["visual.proj"] = bad["visual.proj"][:, :-1]
    with pytest.raises(ValueError):
        load_clip_weights(model, bad)
    del published["ln_final.weight"]
    with pytest.raises(KeyError):
        load_clip_weights(model, published)
    torch.save({"state_dict": {"visual.proj": object()}}, path)          # not tensors only: weights_only refuses it
    with pytest.raises(Exception):
        load_clip_weights(model, path)


def test_transformers_layout_safetensors_file_loads_and_matches_the_oracle_tower(tmp_path):
    """A `transformers.CLIPModel` checkpoint (safetensors, Hugging Face key layout) fills clip_model.*: the converted tensors
    are the ones the independent mapping of the oracle cross-check produces, and the oracle's vision tower on them gives what
    the transformers model itself computes."""
    transformers = pytest.importorskip("transformers")
    from safetensors.torch import save_file
    from otter_amd.clip_weights import load_clip_weights
    from otter_amd.config import ModelConfig, SharedConfig, clip_arch
    from otter_amd.otter import OTTER
    from oracle import clip_ref
    from tests.test_oracle_clip import _hf_to_openai
    arch = clip_arch("tiny-test")
    cfg = transformers.CLIPConfig(
        text_config=dict(hidden_size=arch.transformer_width, intermediate_size=4 * arch.transformer_width, num_hidden_layers=arch.transformer_layers,
                         num_attention_heads=arch.transformer_heads, vocab_size=arch.vocab_size, max_position_embeddings=arch.context_length,
                         hidden_act="quick_gelu", projection_dim=arch.embed_dim),
        vision_config=dict(hidden_size=arch.vision_width, intermediate_size=4 * arch.vision_width, num_hidden_layers=arch.vision_layers,
                           num_attention_heads=arch.vision_heads, image_size=arch.image_resolution, patch_size=arch.vision_patch_size,
                           hidden_act="quick_gelu", projection_dim=arch.embed_dim),
        projection_dim=arch.embed_dim)
    torch.manual_seed(5)
    hf = transformers.CLIPModel(cfg).eval()
    path = str(tmp_path / "model.safetensors")
    save_file({k: v.contiguous() for k, v in hf.state_dict().items()}, path)
    mc = ModelConfig(clip_model="tiny-test", transformer_dim=128, transformer_layers=2, transformer_heads=2, vision_pooling_output_dim=128,
                     text_pooling_output_dim=64, proprio_hidden_dim=64, proprio_output_dim=32, pooling_heads=4)
    model = OTTER(mc, SharedConfig(image_size=112, seq_length=4, action_horizon=4))
    unused = load_clip_weights(model, path)
    assert unused == []
    sd = model.state_dict()
    want = _hf_to_openai(hf.state_dict(), arch.vision_layers, arch.transformer_layers)
    for k, v in want.items():
        assert torch.equal(sd[k], v.float()), k
    assert torch.equal(sd["clip_model.logit_scale"], hf.state_dict()["logit_scale"])
    # and the tensors mean what they should: pooled image embedding of the oracle tower == transformers' own
    px = torch.randn(2, 3, 112, 112, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        ref = hf.get_image_features(pixel_values=px)
        ref = ref if isinstance(ref, torch.Tensor) else ref.pooler_output
    got = clip_ref.encode_image(px, {k: v.float() for k, v in sd.items()})
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("h,w,size", [(180, 320, 224), (720, 1280, 224), (300, 200, 224), (224, 224, 224), (512, 512, 112)])
def test_resize_arithmetic_is_pil_bicubic_bit_for_bit(h, w, size):
    """The fixed-point tables and two-pass arithmetic the GPU resize kernel uses, against PIL itself (down-scaling with
    antialiasing, up-scaling, landscape and portrait crops)."""
    from PIL import Image
    from otter_amd.vision_tf import crop2square, crop_resize_reference_arithmetic, pil_to_uint8_chw
    rng = np.random.default_rng(h * 7 + w)
    frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    frame[: h // 3] = (frame[: h // 3] // 64) * 64                      # flat regions and hard edges as well as noise
    want = pil_to_uint8_chw(crop2square(Image.fromarray(frame)).resize((size, size), Image.Resampling.BICUBIC)).numpy()
    got = crop_resize_reference_arithmetic(frame, size)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_torch_custom_ops_are_registered_and_have_no_cpu_implementation():
    """``torch.ops.otter_hip.*``: one operator per kernel entry point of the path, shape inference on meta tensors for every one
    of them, and a loud failure on CPU tensors."""
    import otter_amd.torch_ops as T
    ns = torch.ops.otter_hip
    assert len(T.OP_NAMES) == 24
    for name in T.OP_NAMES:
        assert hasattr(ns, name), name
    f16, f32, i32, u8, f64 = torch.float16, torch.float32, torch.int32, torch.uint8, torch.float64
    meta = lambda *shape, dtype=f16: torch.empty(*shape, dtype=dtype, device="meta")          # noqa: E731
    y = ns.gemm(meta(300, 128), meta(256, 128), 256, None, 1, False, None)
    assert tuple(y.shape) == (300, 256) and y.dtype == f16
    y = ns.gemm(meta(300, 128), meta(256, 128), 250, None, 0, False, meta(300, 250, dtype=f32))
    assert tuple(y.shape) == (300, 250) and y.dtype == f32
    y = ns.gemm(meta(300, 128), meta(256, 128), 256, meta(256, dtype=f32), 1, False, None, meta(300, 2, dtype=f32), meta(256, dtype=f32))
    assert tuple(y.shape) == (300, 256) and y.dtype == f16
    assert ns.gemm_into(meta(300, 256), meta(300, 128), meta(256, 128), 256, None, 0, None) is None
    assert ns.gemm_residual_(meta(300, 256), meta(300, 128), meta(256, 128), 256, None) is None
    assert tuple(ns.gemm_residual_stats_(meta(300, 256), meta(300, 128), meta(256, 128), 256, None, 1e-5).shape) == (300, 2)
    assert tuple(ns.row_stats(meta(300, 128), 1e-5).shape) == (300, 2)
    assert ns.layernorm(meta(7, 64, dtype=f32), None, None, 1e-5, 0).dtype == torch.bfloat16
    y = ns.im2col_patches(meta(4, 3, 224, 224, dtype=u8), 14, 640, [0.5, 0.5, 0.5], [0.2, 0.2, 0.2], 1)
    assert tuple(y.shape) == (4 * 256, 640) and y.dtype == f16
    y = ns.vit_embed_lnpre(meta(4 * 256, 1024, dtype=f32), meta(1024, dtype=f32), meta(257, 1024, dtype=f32), meta(1024, dtype=f32), meta(1024, dtype=f32),
                           4, 256, 1e-5, 1)
    assert tuple(y.shape) == (4 * 257, 1024) and y.dtype == f16
    y = ns.text_embed(meta(3, 77, dtype=i32), 15, meta(49408, 768, dtype=f32), meta(77, 768, dtype=f32), 2)
    assert tuple(y.shape) == (45, 768) and y.dtype == f32
    assert tuple(ns.attn_fwd_d64(meta(2 * 257, 3 * 1024), 2, 257, 16, False).shape) == (514, 1024)
    assert tuple(ns.attn_policy(meta(3, 12, 3 * 768, dtype=f32), 1, 4, 8).shape) == (3, 768)
    y = ns.attn_policy_rows(meta(2, 3 * 768, dtype=f32), meta(3, 12, 3 * 768, dtype=f32), meta(2, dtype=i32), meta(3, dtype=i32), 8)
    assert tuple(y.shape) == (2, 768)
    assert tuple(ns.pos_emb_pack(meta(196, 512, dtype=f32), 1).shape) == (208 * 512,)
    y = ns.text_aware_pool(meta(4, 257, 768), meta(2, 15, 768, dtype=f32), meta(2, 256 * 768), meta(2, dtype=f32), 2, 2, 1, 256)
    assert tuple(y.shape) == (2, 2, 15, 768) and y.dtype == f32
    assert ns.l2norm_rows(meta(5, 768)).dtype == f32
    blk = [meta(8, dtype=f32) for _ in range(12)]
    assert ns.readout_pool_multi(blk, [1], [meta(4, 64, dtype=f32)], [meta(64, dtype=f32)], [meta(64, dtype=f32)], [meta(30, 128, dtype=f32)], [None],
                                 [meta(2, 256, dtype=f32)], [2], [15], [4], [64], [8], [256]) is None
    assert tuple(ns.linear_f32(meta(5, 10, dtype=f32), meta(10, 256, dtype=f32), None, 3).shape) == (5, 256)
    assert ns.linear_f32_into(meta(5, 256, dtype=f32), meta(5, 10, dtype=f32), meta(10, 256, dtype=f32), None, 0) is None
    assert ns.repeat_rows_into(meta(12, 128, dtype=f32), meta(4, 128, dtype=f32), 3) is None
    assert tuple(ns.convert_rows(meta(24, 768, dtype=f32), 2, 0, 1, meta(2, dtype=i32), 12, -1, -1).shape) == (2, 768)
    assert tuple(ns.convert_rows(meta(4, 700, dtype=f32), 1, 768, 3, None, 1, -1, 0).shape) == (12, 768)
    assert tuple(ns.add_pe(meta(3, 12, 768, dtype=f32), meta(12, 768, dtype=f32), 12, 0).shape) == (36, 768)
    assert tuple(ns.add_pe_rows(meta(3, 12, 768, dtype=f32), meta(12, 768, dtype=f32), meta(2, dtype=i32), meta(3, dtype=i32)).shape) == (2, 768)
    assert ns.window_append(meta(3, 12, 768, dtype=f32), meta(3, 768, dtype=f32), meta(3, 12, 768, dtype=f32), meta(3, dtype=i32), meta(3, dtype=i32)) is None
    r, a10, ab = ns.action_step(meta(5, 12, 10, dtype=f32), meta(5, 10, dtype=f64), meta(5, 4, 4, 10, dtype=f64), meta(5, dtype=i32), 4, False, 0.05, True)
    assert tuple(r.shape) == (5, 7) and tuple(a10.shape) == (5, 10) and tuple(ab.shape) == (5, 12, 10) and r.dtype == f64
    with pytest.raises(NotImplementedError):
        ns.l2norm_rows(torch.randn(3, 8))
    with pytest.raises(NotImplementedError):
        ns.gemm(torch.randn(4, 32), torch.randn(256, 32), 8, None, 0, False, None)


def test_model_code_launches_kernels_through_the_custom_operators():
    """device_model / engine / pooling call the kernels through ``torch.ops.otter_hip`` (the `kernels` namespace), not through
    the ctypes wrappers directly; host-side helpers pass through."""
    import otter_amd.device_model as dm
    import otter_amd.engine as eng
    import otter_amd.pooling as pool
    from otter_amd import ops as raw
    from otter_amd.torch_ops import OP_NAMES, kernels
    assert dm.ops is kernels and eng.ops is kernels and pool.ops is kernels
    assert kernels.pack_weight is raw.pack_weight and kernels.ACT_GELU == raw.ACT_GELU and kernels.OtterHipError is raw.OtterHipError
    assert set(kernels._op) == set(OP_NAMES)
    with pytest.raises(NotImplementedError):                      # a CPU tensor reaches the operator, which has no CPU kernel
        kernels.l2norm_rows(torch.randn(3, 8))
