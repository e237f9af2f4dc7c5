"""Checkpoint files in the reference's layout (``otter/util/engine.py:112-122``, ``otter_interface.py:36-48``):
``<dir>/run.yaml`` with the experiment configuration and ``<dir>/checkpoint_<step>.pt`` holding the model's
``state_dict`` (fp32 tensors, reference key names), so either implementation can load what the other wrote."""
from __future__ import annotations

import os
from typing import Dict

import torch

from .config import ExperimentConfig, dump_run_yaml


def save_checkpoint(model, args: ExperimentConfig, out_dir: str, step: int) -> str:
    os.makedirs(out_dir, exist_ok=True)
    yaml_path = os.path.join(out_dir, "run.yaml")
    if not os.path.exists(yaml_path):
        dump_run_yaml(args, yaml_path)
    target = model.module if hasattr(model, "module") else model
    state: Dict[str, torch.Tensor] = {k: v.detach().to("cpu").contiguous().clone() for k, v in target.state_dict().items()}
    path = os.path.join(out_dir, f"checkpoint_{step}.pt")
    tmp = path + ".tmp"
    torch.save(state, tmp)
    os.replace(tmp, path)                                   # a reader never sees a half-written file
    return path


def load_checkpoint_state(path: str) -> Dict[str, torch.Tensor]:
    """Tensors only (``weights_only=True``: nothing in the file is executed); a DistributedDataParallel ``module.``
    prefix is stripped (``otter/util/misc.py:28-50``)."""
    state = torch.load(path, map_location="cpu", weights_only=True)
    if any(k.startswith("module.") for k in state):
        state = {k.removeprefix("module."): v for k, v in state.items()}
    return state
