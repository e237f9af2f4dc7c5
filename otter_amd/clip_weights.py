"""Loading published CLIP weights into an :class:`otter_amd.otter.OTTER` (``clip.load(name)`` at ``otter/policy/otter.py:62``).

The reference downloads the OpenAI archive by name; here the caller supplies a file (there is no network in the
deployment this package targets).  Accepted: a ``state_dict`` of TENSORS ONLY, fp16 / bf16 / fp32, as a ``torch.save`` file
(optionally wrapped as ``{"state_dict": ...}``), as a ``.safetensors`` file, or as a dict in memory, in one of two key layouts:

* the OpenAI layout (``visual.conv1.weight``, ``transformer.resblocks.0.attn.in_proj_weight``, ``token_embedding.weight`` ...),
  which is also what OpenCLIP's ``ViT-L-14`` checkpoints use (``open_clip_pytorch_model.bin`` / ``open_clip_model.safetensors``
  of ``datacomp_xl_s13b_b90k``, ``script/open_clip_visualization.py:25-38``; their ``attn_mask`` buffer is reported unused);
* the Hugging Face ``transformers.CLIPModel`` layout (``vision_model.encoder.layers.0.self_attn.q_proj.weight`` ...:
  ``model.safetensors`` / ``pytorch_model.bin`` of ``openai/clip-vit-large-patch14``), converted on the fly: q / k / v
  projections concatenated into ``in_proj``, the two projection matrices transposed.

``torch.save`` files are read with ``torch.load(weights_only=True)`` and safetensors files hold raw tensor bytes: nothing in
either is executed.  The OpenAI download itself (``ViT-L-14.pt``, what ``clip.load`` fetches) is a TorchScript archive: a zip
of ``data.pkl`` (the pickled module tree), ``code/`` (TorchScript source) and ``data/<n>`` (raw storages).  It is read here
WITHOUT ``torch.jit.load`` and without running anything from the file (:func:`read_torchscript_archive`): the pickle is walked
by an unpickler that admits tensor re-assembly and inert containers only, the ``code/`` entries are never opened.
An OTTER checkpoint needs none of this: it already contains the CLIP tensors.
"""
from __future__ import annotations

import io
import pickle
import zipfile
from collections import OrderedDict
from typing import Dict, List, Union

import torch

# entries of the OpenAI archive that are not tensors of the model
_NOT_PARAMETERS = ("input_resolution", "context_length", "vocab_size")


# ------------------------------------------------------------------------------------------- TorchScript archive, no code run
_STORAGE_DTYPES = {"HalfStorage": torch.float16, "FloatStorage": torch.float32, "DoubleStorage": torch.float64,
                   "BFloat16Storage": torch.bfloat16, "LongStorage": torch.int64, "IntStorage": torch.int32,
                   "ShortStorage": torch.int16, "CharStorage": torch.int8, "ByteStorage": torch.uint8, "BoolStorage": torch.bool}


class _StorageType:
    """Stands for ``torch.<X>Storage`` inside the pickle: carries the element type, constructs nothing."""

    def __init__(self, name: str):
        self.dtype = _STORAGE_DTYPES[name]


class _ScriptObject:
    """Stands for any ``__torch__.*`` class (a scripted module or one of its sub-objects): keeps the attribute dict, runs no code."""

    def __init__(self):
        self.state = {}

    def __setstate__(self, state):
        self.state = state if isinstance(state, dict) else {"__state__": state}


def _rebuild_tensor(storage: torch.Tensor, offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    size, stride = tuple(int(x) for x in size), tuple(int(x) for x in stride)
    need = int(offset) + 1 + sum((n - 1) * st for n, st in zip(size, stride)) if all(n > 0 for n in size) else 0
    if any(n < 0 for n in size) or any(st < 0 for st in stride) or int(offset) < 0 or need > storage.numel():
        raise pickle.UnpicklingError(f"tensor view (offset {offset}, size {size}, stride {stride}) outside its storage of {storage.numel()}")
    return storage.as_strided(size, stride, int(offset)).clone()


def _identity(x, *_):
    return x


class _TensorsOnlyUnpickler(pickle.Unpickler):
    """Unpickler for the ``data.pkl`` / ``constants.pkl`` of a TorchScript archive.  ``find_class`` resolves exactly:
    tensor re-assembly (re-implemented above), storage type markers, ``OrderedDict``, the list builders TorchScript emits, and
    ``__torch__.*`` names (to an inert attribute holder).  Everything else -- hence any callable from the file -- is refused."""

    def __init__(self, data: bytes, read_storage):
        super().__init__(io.BytesIO(data))
        self._read_storage = read_storage

    def find_class(self, module: str, name: str):
        if module.startswith("__torch__"):
            return _ScriptObject
        if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
            return _rebuild_tensor
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _identity
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageType(name)
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module == "torch.jit._pickle" and name in ("build_intlist", "build_tensorlist", "build_doublelist", "build_boollist", "restore_type_tag"):
            return _identity
        raise pickle.UnpicklingError(f"refusing to resolve {module}.{name} from a TorchScript archive (tensors and containers only)")

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], _StorageType)):
            raise pickle.UnpicklingError(f"unexpected persistent id {pid!r}")
        _, stype, key, _location, numel = pid[:5]
        raw = self._read_storage(str(key))
        if len(raw) < int(numel) * torch.empty((), dtype=stype.dtype).element_size():
            raise pickle.UnpicklingError(f"storage {key} holds {len(raw)} bytes, fewer than its {numel} elements need")
        return torch.frombuffer(bytearray(raw), dtype=stype.dtype, count=int(numel)) if int(numel) else torch.empty(0, dtype=stype.dtype)


def is_torchscript_archive(path: str) -> bool:
    """A zip whose entries include ``<name>/constants.pkl`` or ``<name>/code/``: written by ``torch.jit.save``."""
    if not zipfile.is_zipfile(path):
        return False
    with zipfile.ZipFile(path) as z:
        return any(n.endswith("/constants.pkl") or "/code/" in n for n in z.namelist())


def read_torchscript_archive(path: str) -> Dict[str, torch.Tensor]:
    """Tensors of a ``torch.jit.save`` archive as a flat ``{dotted.attribute.path: tensor}`` dict (what
    ``torch.jit.load(path).state_dict()`` holds, plus tensor-valued plain attributes), read without executing anything."""
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
        pkl = [n for n in names if n.endswith("/data.pkl")]
        if len(pkl) != 1:
            raise ValueError(f"{path}: expected one data.pkl, found {pkl}")
        root = pkl[0][: -len("data.pkl")]
        obj = _TensorsOnlyUnpickler(z.read(pkl[0]), lambda key: z.read(root + "data/" + key)).load()
    out: Dict[str, torch.Tensor] = {}

    def walk(o, prefix: str):
        if isinstance(o, torch.Tensor):
            out[prefix[:-1]] = o
        elif isinstance(o, _ScriptObject):
            for k, v in o.state.items():
                if isinstance(k, str):
                    walk(v, prefix + k + ".")

    walk(obj, "")
    if not out:
        raise ValueError(f"{path}: no tensors found in the archive's module tree")
    return out


def clip_state_dict_from_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path, device="cpu")
    elif is_torchscript_archive(path):
        sd = read_torchscript_archive(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)   # raises on anything but plain tensors / containers
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}


def is_transformers_layout(sd: Dict[str, torch.Tensor]) -> bool:
    return any(k.startswith(("vision_model.", "text_model.")) for k in sd)


def from_transformers_layout(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``transformers.CLIPModel.state_dict()`` -> the OpenAI key layout.  Same tensors: the attention's three projections
    are concatenated in q, k, v order (``nn.MultiheadAttention.in_proj_*``), ``visual_projection`` / ``text_projection``
    are ``nn.Linear`` weights there and plain matrices multiplied from the right here (transposed); buffers such as
    ``position_ids`` have no counterpart and are dropped."""
    out: Dict[str, torch.Tensor] = {}
    fixed = {
        "vision_model.embeddings.class_embedding": "visual.class_embedding",
        "vision_model.embeddings.patch_embedding.weight": "visual.conv1.weight",
        "vision_model.embeddings.position_embedding.weight": "visual.positional_embedding",
        "vision_model.pre_layrnorm.weight": "visual.ln_pre.weight",              # (sic: the upstream attribute name)
        "vision_model.pre_layrnorm.bias": "visual.ln_pre.bias",
        "vision_model.post_layernorm.weight": "visual.ln_post.weight",
        "vision_model.post_layernorm.bias": "visual.ln_post.bias",
        "text_model.embeddings.token_embedding.weight": "token_embedding.weight",
        "text_model.embeddings.position_embedding.weight": "positional_embedding",
        "text_model.final_layer_norm.weight": "ln_final.weight",
        "text_model.final_layer_norm.bias": "ln_final.bias",
        "logit_scale": "logit_scale",
    }
    for src, dst in fixed.items():
        if src in sd:
            out[dst] = sd[src]
    for src, dst in (("visual_projection.weight", "visual.proj"), ("text_projection.weight", "text_projection")):
        if src in sd:
            out[dst] = sd[src].t().contiguous()
    per_block = (("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"), ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj"),
                 ("self_attn.out_proj", "attn.out_proj"))
    for tower, root in (("vision_model", "visual.transformer"), ("text_model", "transformer")):
        i = 0
        while f"{tower}.encoder.layers.{i}.self_attn.q_proj.weight" in sd:
            h, o = f"{tower}.encoder.layers.{i}.", f"{root}.resblocks.{i}."
            for kind in ("weight", "bias"):
                out[o + "attn.in_proj_" + kind] = torch.cat([sd[h + f"self_attn.{x}_proj.{kind}"] for x in "qkv"], 0)
                for a, b in per_block:
                    out[o + b + "." + kind] = sd[h + a + "." + kind]
            i += 1
    return out


def load_clip_weights(model, source: Union[str, Dict[str, torch.Tensor]]) -> List[str]:
    """Copy CLIP tensors into ``model.clip_model.*`` (converted to the parameter's dtype).  Every CLIP tensor of the model
    must be present with the right shape; returns the keys of ``source`` that were not used."""
    sd = clip_state_dict_from_file(source) if isinstance(source, str) else dict(source)
    if is_transformers_layout(sd):
        sd = from_transformers_layout(sd)
    own = model.state_dict()
    used, unused = set(), []
    with torch.no_grad():
        for k, v in sd.items():
            key = k if k.startswith("clip_model.") else "clip_model." + k
            if key not in own or k in _NOT_PARAMETERS:
                unused.append(k)
                continue
            if tuple(own[key].shape) != tuple(v.shape):
                raise ValueError(f"{k}: shape {tuple(v.shape)} in the file, {tuple(own[key].shape)} in {model.clip_model_name}")
            own[key].copy_(v.to(own[key].dtype))
            used.add(key)
    missing = [k for k in own if k.startswith("clip_model.") and k not in used]
    if missing:
        raise KeyError(f"{len(missing)} CLIP tensors missing from the source, e.g. {missing[:3]}")
    model._clip_dev = None                              # packed device copies are stale
    model._dev = None
    model._rollout = None
    return unused
