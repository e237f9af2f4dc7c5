"""Rollout interface: ``OtterInference`` / ``OtterInferenceSim`` (``otter/policy/otter_interface.py``).

Same constructor, ``__call__`` and ``reset`` as the reference classes; the model runs on the MI355X.
Per call the host does what the reference's host code does (pose -> 10-d proprio, BPE, crop + bicubic resize,
delta -> absolute decode); pixels are uploaded as uint8 and normalised inside the patch-extraction kernel,
the text tower runs once per distinct prompt, and the history window / K/V cache stay on the GPU.
"""
from __future__ import annotations

import os
from typing import Dict, Union

import numpy as np
import torch
from PIL import Image
from scipy.spatial.transform import Rotation

from .action_utils import convert_abs_action
from .config import ExperimentConfig, load_run_yaml
from .otter import OTTER, create_text_mask
from .tokenizer import tokenize
from .vision_tf import crop2square, crop_resize_device, pil_to_uint8_chw


def load_state_dict_flexible(model, state_dict):
    """Strict load that accepts checkpoints saved from a DistributedDataParallel wrapper
    (``otter/util/misc.py:28-50``): a ``module.`` prefix is stripped."""
    if any(k.startswith("module.") for k in state_dict):
        state_dict = {k.removeprefix("module."): v for k, v in state_dict.items()}
    target = model.module if hasattr(model, "module") else model
    target.load_state_dict(state_dict)
    return model


class _InferenceBase:
    def __init__(self, model_ckpt_folder: str, ckpt_id: int, device: str = "cuda") -> None:
        run_yaml = os.path.join(model_ckpt_folder, "run.yaml")
        ckpt = os.path.join(model_ckpt_folder, f"checkpoint_{ckpt_id}.pt")
        self.args: ExperimentConfig = load_run_yaml(run_yaml)
        self.device = device
        self.model = OTTER(model_config=self.args.model_cfg, shared_config=self.args.shared_cfg)
        print(f"Loading model from checkpoint: {ckpt}")
        self.model = load_state_dict_flexible(self.model, torch.load(ckpt, map_location="cpu", weights_only=True))
        for p in self.model.parameters():
            p.requires_grad = False
        self.model = self.model.to(device)
        self.model.eval()
        print(self.model)

    def _tokens_and_mask(self, text: str):
        ids = tokenize([text]).squeeze(0)                       # (77,) int32, host
        mask = None
        if self.model.pool_true_text:
            mask = create_text_mask(ids, self.model.sot_token, self.model.eot_token, self.model.first_k_tokens)
        return ids, mask

    def _decode_to_numpy(self, actions: Union[np.ndarray, torch.Tensor]) -> np.ndarray:
        if isinstance(actions, torch.Tensor):
            actions = actions.detach().cpu().numpy()
        return actions


class OtterInference(_InferenceBase):
    def __init__(self, model_ckpt_folder: str, ckpt_id: int, device: str = "cuda") -> None:
        super().__init__(model_ckpt_folder, ckpt_id, device)
        self.current_proprio = None       # (proprio_input_dim,) float64, set by _proprocess_proprio

    def _proprocess_proprio(self, proprio: np.ndarray, gripper: np.ndarray) -> torch.Tensor:
        """[x, y, z, roll, pitch, yaw] + [gripper] -> [xyz, first two rows of R("XYZ"), clip(gripper, 0, 1)]
        (``otter_interface.py:60-84``)."""
        rot = Rotation.from_euler("XYZ", proprio[3:], degrees=False).as_matrix()[:2, :].reshape((6,))
        data = np.concatenate((np.concatenate([proprio[:3], rot], axis=-1), np.clip(gripper, 0, 1)), axis=-1)
        self.current_proprio = data.copy()
        return torch.tensor(data).float()

    def _preprocess_images(self, images: Dict[str, Image.Image]) -> torch.Tensor:
        """Centre-crop, bicubic resize to image_size (``otter_interface.py:86-111``) -> uint8 (cams, 3, S, S).

        The raw frames are uploaded and cropped / resized on the GPU by a kernel pair that reproduces PIL's 8-bit bicubic
        resampler bit for bit; on a CPU-resident model the host does it with PIL as upstream."""
        assert set(images.keys()) == set(self.model.camera_keys)
        size = self.args.shared_cfg.image_size
        device = self.model.input_projection.weight.device
        if device.type != "cuda":
            return torch.stack([pil_to_uint8_chw(crop2square(images[k]).resize((size, size), Image.Resampling.BICUBIC))
                                for k in self.model.camera_keys], dim=0)
        out = []
        for k in self.model.camera_keys:
            frame = torch.from_numpy(np.asarray(images[k].convert("RGB"), dtype=np.uint8).copy())
            out.append(crop_resize_device(frame[None].to(device), size)[0])
        return torch.stack(out, dim=0)

    def _decode_actions(self, actions: Union[np.ndarray, torch.Tensor]) -> np.ndarray:
        """Delta chunk (T, D) -> absolute chunk relative to the last proprio (``otter_interface.py:113-133``)."""
        actions = self._decode_to_numpy(actions)
        return convert_abs_action(actions[None, ...], self.current_proprio[None, None, ...])[0]

    def reset(self):
        self.current_proprio = None
        self.model.reset_cache()

    def __call__(self, images: Dict[str, Image.Image], text: str, proprio: np.ndarray, gripper: np.ndarray):
        prop = self._proprocess_proprio(proprio, gripper)
        ids, mask = self._tokens_and_mask(text)
        imgs = self._preprocess_images(images)
        action = self.model.forward_inference(images=imgs, text=ids, proprio=prop, text_mask=mask)
        return self._decode_actions(action)


class OtterInferenceSim(_InferenceBase):
    """Simulator variant (``otter_interface.py:188-307``): proprio passed through, images only ``to_tensor``-ed
    (no crop / resize), actions returned as predicted (no absolute decode)."""

    def _proprocess_proprio(self, proprio: np.ndarray) -> torch.Tensor:
        return torch.tensor(proprio).float()

    def _preprocess_images(self, images: Dict[str, Image.Image]) -> torch.Tensor:
        assert set(images.keys()) == set(self.model.camera_keys)
        return torch.stack([pil_to_uint8_chw(images[k]) for k in self.model.camera_keys], dim=0)

    def _decode_actions(self, actions):
        return self._decode_to_numpy(actions)

    def reset(self):
        self.model.reset_cache()

    def __call__(self, images: Dict[str, Image.Image], text: str, proprio: np.ndarray):
        prop = self._proprocess_proprio(proprio)
        ids, mask = self._tokens_and_mask(text)
        imgs = self._preprocess_images(images)
        action = self.model.forward_inference(images=imgs, text=ids, proprio=prop, text_mask=mask)
        return self._decode_actions(action)
