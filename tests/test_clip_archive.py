"""The OpenAI CLIP download is a TorchScript archive (``clip.load`` at ``otter/policy/otter.py:72`` fetches ``ViT-L-14.pt``).
`clip_weights.read_torchscript_archive` reads its tensors without ``torch.jit.load`` and without running anything from the file.
No published archive exists offline, so the archives read here are written in the test from small OpenAI-structured modules
(scripted and traced, fp16 and fp32) -- the format is the writer's (`torch.jit.save`), the reader is this package's."""
import os
import pickle
import zipfile

import pytest
import torch
import torch.nn as nn

from otter_amd import clip_weights as cw


class _Block(nn.Module):
    def __init__(self, width: int, heads: int):
        super().__init__()
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_1 = nn.LayerNorm(width)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(width, 4 * width))
        self.mlp.add_module("c_proj", nn.Linear(4 * width, width))
        self.ln_2 = nn.LayerNorm(width)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.ln_1(x)
        x = x + self.attn(y, y, y, need_weights=False)[0]
        return x + self.mlp(self.ln_2(x))


class _Tower(nn.Module):
    def __init__(self, width: int, heads: int, layers: int):
        super().__init__()
        self.resblocks = nn.Sequential(*[_Block(width, heads) for _ in range(layers)])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.resblocks(x)


class _MiniClip(nn.Module):
    """OpenAI key layout in miniature: parameters on the root, a transformer of resblocks, plain tensor attributes."""

    def __init__(self):
        super().__init__()
        self.transformer = _Tower(16, 2, 2)
        self.positional_embedding = nn.Parameter(torch.randn(5, 16))
        self.text_projection = nn.Parameter(torch.randn(16, 8))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.5)
        self.token_embedding = nn.Embedding(32, 16)
        self.ln_final = nn.LayerNorm(16)
        self.register_buffer("input_resolution", torch.tensor(224))

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        x = self.token_embedding(ids) + self.positional_embedding          # (B, 5, 16)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        return self.ln_final(x) @ self.text_projection * self.logit_scale.exp()


@pytest.mark.parametrize("kind,half", [("script", False), ("script", True), ("trace", False)])
def test_torchscript_archive_read_without_running_it(tmp_path, kind, half):
    torch.manual_seed(0)
    m = _MiniClip().eval()
    if half:
        m = m.half()
    ids = torch.randint(0, 32, (3, 5))
    sm = torch.jit.script(m) if kind == "script" else torch.jit.trace(m, ids)
    path = str(tmp_path / "mini.pt")
    torch.jit.save(sm, path)
    assert cw.is_torchscript_archive(path)
    got = cw.clip_state_dict_from_file(path)                   # the entry point OTTER users call, routed by file content
    ref = sm.state_dict()
    assert set(ref) <= set(got), sorted(set(ref) - set(got))
    for k, v in ref.items():
        assert got[k].dtype == v.dtype and tuple(got[k].shape) == tuple(v.shape) and torch.equal(got[k], v), k
    assert "transformer.resblocks.1.attn.in_proj_weight" in got and "transformer.resblocks.0.mlp.c_fc.bias" in got
    assert got["logit_scale"].dim() == 0 and int(got["input_resolution"]) == 224


def test_plain_checkpoints_are_not_mistaken_for_archives(tmp_path):
    sd = {"a": torch.randn(3, 4), "b": torch.arange(5)}
    p = str(tmp_path / "plain.pt")
    torch.save(sd, p)
    assert not cw.is_torchscript_archive(p)
    got = cw.clip_state_dict_from_file(p)
    assert torch.equal(got["a"], sd["a"]) and torch.equal(got["b"], sd["b"])
    assert not cw.is_torchscript_archive(__file__)


class _Touch:
    """Pickles to a call of ``os.system``: what a hostile archive would carry."""

    def __init__(self, target):
        self.target = target

    def __reduce__(self):
        return (os.system, (f"touch {self.target}",))


def test_archive_with_a_callable_in_its_pickle_is_refused(tmp_path):
    """A data.pkl that names anything but tensor re-assembly and containers is refused, and nothing it names runs."""
    marker = tmp_path / "ran"
    path = str(tmp_path / "evil.pt")
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("evil/data.pkl", pickle.dumps({"w": _Touch(str(marker))}, protocol=2))
        z.writestr("evil/constants.pkl", pickle.dumps((), protocol=2))
        z.writestr("evil/code/__torch__.py", "")
        z.writestr("evil/version", "3\n")
    assert cw.is_torchscript_archive(path)
    with pytest.raises(pickle.UnpicklingError, match="refusing to resolve"):
        cw.read_torchscript_archive(path)
    assert not marker.exists()


def test_tensor_view_outside_its_storage_is_refused(tmp_path):
    """Size / stride / offset come from the file: a view that reaches past its storage is an error, not a read."""
    storage = torch.arange(8, dtype=torch.float32)
    with pytest.raises(pickle.UnpicklingError, match="outside its storage"):
        cw._rebuild_tensor(storage, 4, (2, 4), (4, 1))
    with pytest.raises(pickle.UnpicklingError, match="outside its storage"):
        cw._rebuild_tensor(storage, 0, (3,), (-1,))
    assert torch.equal(cw._rebuild_tensor(storage, 2, (2, 3), (3, 1)), storage[2:8].view(2, 3))


def test_load_clip_weights_from_an_archive_into_otter(tmp_path):
    """End to end: a scripted module holding a full set of (tiny) CLIP tensors in the OpenAI layout -> archive -> OTTER."""
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.otter import OTTER
    from otter_amd.params import otter_specs, random_state_dict
    mc = ModelConfig(clip_model="tiny-test", transformer_dim=128, transformer_layers=1, transformer_heads=2, vision_pooling_output_dim=128,
                     text_pooling_output_dim=64, proprio_hidden_dim=64, proprio_output_dim=32, pooling_heads=4)
    sc = SharedConfig(image_size=112, seq_length=2, action_horizon=4)
    sd = random_state_dict(otter_specs(mc, sc), seed=4)
    clip_sd = {k[len("clip_model."):]: v.half() for k, v in sd.items() if k.startswith("clip_model.")}

    class Holder(nn.Module):                     # a module tree with the OpenAI names; scripting stores every tensor
        def forward(self, x: torch.Tensor) -> torch.Tensor:
            return x

    root = Holder()
    for key, v in clip_sd.items():
        node = root
        parts = key.split(".")
        for name in parts[:-1]:
            if name not in node._modules:
                node.add_module(name, Holder())
            node = node._modules[name]
        node.register_parameter(parts[-1], nn.Parameter(v.clone(), requires_grad=False))
    path = str(tmp_path / "tiny-clip.pt")
    torch.jit.save(torch.jit.script(root), path)
    model = OTTER(mc, sc)
    unused = cw.load_clip_weights(model, path)
    assert unused == []
    own = model.state_dict()
    for k, v in clip_sd.items():
        assert torch.equal(own["clip_model." + k], v.float()), k
