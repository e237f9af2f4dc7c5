"""Configuration objects of the hot path and a tolerant ``run.yaml`` reader.

Field names and defaults of :class:`ModelConfig` / :class:`SharedConfig` are the
checkpoint contract of the reference (``otter/util/args.py:121-178`` and
``:219-264``): a ``run.yaml`` written by the reference's trainer is a python-object
dump of its ``ExperimentConfig`` (``script/train.py:229-231``) and is read back
with ``yaml.Loader`` (``otter/policy/otter_interface.py:33``).  Loading that dump
with the stock loader imports the reference's TensorFlow data layer, so
:func:`load_run_yaml` maps the object tags onto the classes below and turns every
unknown ``python/name`` / ``python/object`` tag into inert data instead.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import yaml


@dataclass
class ModelConfig:
    clip_model: str = "ViT-L/14"
    num_readouts: int = 4
    proprio_input_dim: int = 10
    proprio_hidden_dim: int = 256
    proprio_output_dim: int = 64
    text_pooling_output_dim: int = 128
    first_k_tokens: int = 15
    vision_pooling_output_dim: int = 512
    pooling_heads: int = 8
    pooling_layers: int = 2
    action_dim: int = 10
    max_position_embeddings: int = 32
    transformer_layers: int = 8
    transformer_heads: int = 8
    transformer_dim: int = 768
    transformer_expansion_factor: int = 1
    attention_probs_dropout_prob: float = 0.0
    hidden_dropout_prob: float = 0.0
    pool_true_text: bool = False


@dataclass
class SharedConfig:
    batch_size: int = 64
    rot_6d: bool = True
    seq_length: int = 12
    seed: int = 0
    start_epoch: int = 0
    save_every: int = 5000
    val_every: int = 1000
    num_val_steps: int = 50
    resume: Optional[str] = None
    camera_keys: Tuple[str, ...] = ("image_primary", "image_wrist")
    mse_loss: bool = False
    action_horizon: int = 12
    use_delta_action: bool = True
    scale_action: Optional[str] = None
    image_size: int = 224


@dataclass
class OptimizerConfig:
    weight_decay: float = 0.01
    lr: Optional[float] = 3e-4
    min_lr: float = 0.0


@dataclass
class ExperimentConfig:
    """Only the parts of the reference's experiment tree the hot path reads."""
    model_cfg: ModelConfig = field(default_factory=ModelConfig)
    shared_cfg: SharedConfig = field(default_factory=SharedConfig)
    optimizer_cfg: OptimizerConfig = field(default_factory=OptimizerConfig)
    extras: Dict[str, Any] = field(default_factory=dict)


@dataclass(frozen=True)
class ClipArch:
    """Hyper-parameters of a CLIP checkpoint (OpenAI ``CLIP.__init__`` argument order)."""
    embed_dim: int
    image_resolution: int
    vision_layers: int
    vision_width: int
    vision_patch_size: int
    context_length: int
    vocab_size: int
    transformer_width: int
    transformer_heads: int
    transformer_layers: int
    activation: str = "quick_gelu"   # "gelu" for OpenCLIP datacomp ViT-L/14 weights

    @property
    def vision_heads(self) -> int:
        return self.vision_width // 64

    @property
    def grid(self) -> int:
        return self.image_resolution // self.vision_patch_size


CLIP_ARCHS: Dict[str, ClipArch] = {
    "ViT-L/14": ClipArch(768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12),
    "ViT-L/14@336px": ClipArch(768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12),
    "ViT-B/16": ClipArch(512, 224, 12, 768, 16, 77, 49408, 512, 8, 12),
    "ViT-B/32": ClipArch(512, 224, 12, 768, 32, 77, 49408, 512, 8, 12),
    # OpenCLIP ViT-L-14 / datacomp_xl_s13b_b90k: same tensors, exact GELU
    # (script/open_clip_visualization.py:25-38).
    "open_clip:ViT-L-14": ClipArch(768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12, "gelu"),
    # Small architecture for fast tests (head dim stays 64 like every CLIP ViT).
    "tiny-test": ClipArch(128, 112, 2, 128, 14, 77, 49408, 64, 1, 2),
}


def clip_arch(name: str) -> ClipArch:
    try:
        return CLIP_ARCHS[name]
    except KeyError:
        raise ValueError(f"unknown CLIP architecture {name!r}; known: {sorted(CLIP_ARCHS)}") from None


# ----------------------------------------------------------------------------
# run.yaml
# ----------------------------------------------------------------------------

class _Bag(dict):
    """Inert stand-in for python objects this package has no class for."""


class _RunYamlLoader(yaml.SafeLoader):
    pass


_KNOWN = {
    "ModelConfig": ModelConfig,
    "SharedConfig": SharedConfig,
    "OptimizerConfig": OptimizerConfig,
}


def _construct_object(loader: _RunYamlLoader, suffix: str, node: yaml.Node):
    cls_name = suffix.rsplit(".", 1)[-1]
    if isinstance(node, yaml.MappingNode):
        data = loader.construct_mapping(node, deep=True)
    elif isinstance(node, yaml.SequenceNode):
        return loader.construct_sequence(node, deep=True)
    else:
        return loader.construct_scalar(node)
    cls = _KNOWN.get(cls_name)
    if cls is None:
        bag = _Bag(data)
        bag["__class__"] = suffix
        return bag
    names = {f.name for f in dataclasses.fields(cls)}
    obj = cls(**{k: v for k, v in data.items() if k in names})
    if isinstance(obj, SharedConfig):
        obj.camera_keys = tuple(obj.camera_keys)
    return obj


def _construct_name(loader: _RunYamlLoader, suffix: str, node: yaml.Node):
    return suffix   # e.g. a function reference of the data layer: keep its dotted name


def _construct_tuple(loader: _RunYamlLoader, node: yaml.Node):
    return tuple(loader.construct_sequence(node, deep=True))


_RunYamlLoader.add_multi_constructor("tag:yaml.org,2002:python/object:", _construct_object)
_RunYamlLoader.add_multi_constructor("tag:yaml.org,2002:python/object/apply:", _construct_object)
_RunYamlLoader.add_multi_constructor("tag:yaml.org,2002:python/object/new:", _construct_object)
_RunYamlLoader.add_multi_constructor("tag:yaml.org,2002:python/name:", _construct_name)
_RunYamlLoader.add_constructor("tag:yaml.org,2002:python/tuple", _construct_tuple)


def load_run_yaml(path) -> ExperimentConfig:
    """Read a ``run.yaml`` written by the reference's trainer (or by :func:`dump_run_yaml`)."""
    raw = yaml.load(Path(path).read_text(), Loader=_RunYamlLoader)
    if not isinstance(raw, dict):
        raise ValueError(f"{path}: expected a mapping at top level, got {type(raw).__name__}")
    cfg = ExperimentConfig()
    for key in ("model_cfg", "shared_cfg", "optimizer_cfg"):
        val = raw.get(key)
        cls = type(getattr(cfg, key))
        if isinstance(val, cls):
            setattr(cfg, key, val)
        elif isinstance(val, dict):
            names = {f.name for f in dataclasses.fields(cls)}
            setattr(cfg, key, cls(**{k: v for k, v in val.items() if k in names}))
        elif val is not None:
            raise ValueError(f"{path}: cannot interpret {key!r} of type {type(val).__name__}")
    cfg.shared_cfg.camera_keys = tuple(cfg.shared_cfg.camera_keys)
    cfg.extras = {k: v for k, v in raw.items() if k not in ("model_cfg", "shared_cfg", "optimizer_cfg")}
    return cfg


def dump_run_yaml(cfg: ExperimentConfig, path) -> None:
    """Write a ``run.yaml`` with the reference's object tags so either side can read it."""
    def tagged(obj, name):
        body = dataclasses.asdict(obj)
        if "camera_keys" in body:
            body["camera_keys"] = list(body["camera_keys"])
        lines = yaml.safe_dump(body, default_flow_style=False, sort_keys=False).splitlines()
        return [f"{name}: !!python/object:otter.util.args.{type(obj).__name__}"] + ["  " + l for l in lines]
    out = ["!!python/object:otter.util.args.ExperimentConfig"]
    out += tagged(cfg.model_cfg, "model_cfg")
    out += tagged(cfg.shared_cfg, "shared_cfg")
    out += tagged(cfg.optimizer_cfg, "optimizer_cfg")
    Path(path).write_text("\n".join(out) + "\n")
