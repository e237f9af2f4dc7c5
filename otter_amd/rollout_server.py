"""Action post-processing of the robot rollout loop (``script/droid_inference.py:97-200``): which element of the
predicted chunks is executed at each control step.

Two modes, as upstream: receding-horizon control (execute the first ``action_exec_horizon`` actions of a chunk, then
predict again) and temporal ensembling (every step predicts a chunk; the action executed now is the
exponentially weighted mean of what the last ``action_exec_horizon`` chunks predicted for this step, weights
``exp(k * i)`` with i = 0 for the oldest chunk, k = 0.05).  :class:`ActionQueue` / :class:`RolloutPolicy` are the
single-episode host form (numpy, as upstream); :class:`DeviceActionQueue` / :class:`BatchedRolloutPolicy` do the same for
many concurrent episodes in one kernel launch per control step, chunks and queue state never leaving the GPU.
"""
from __future__ import annotations

from collections import deque
from typing import Deque, Optional

import numpy as np

from .action_utils import action_10d_to_7d


class ActionQueue:
    def __init__(self, action_exec_horizon: int, receding_horizon_control: bool = False, k: float = 0.05):
        if action_exec_horizon < 1:
            raise ValueError("action_exec_horizon must be at least 1")
        self.horizon, self.receding, self.k = int(action_exec_horizon), bool(receding_horizon_control), float(k)
        self._queue: Deque = deque([], maxlen=self.horizon)

    def clear(self) -> None:
        self._queue.clear()

    def needs_prediction(self) -> bool:
        """Receding-horizon control only predicts when its queue ran empty; ensembling predicts every step."""
        return (not self.receding) or len(self._queue) == 0

    def next_action(self, chunk: Optional[np.ndarray]) -> np.ndarray:
        """``chunk`` (T, A): the prediction of this control step (may be None in receding-horizon mode while queued
        actions remain).  Returns the (A,) action to execute now."""
        if self.receding:
            if len(self._queue) == 0:
                if chunk is None:
                    raise ValueError("the action queue is empty: a new chunk is needed")
                self._queue = deque(np.asarray(chunk)[: self.horizon])
            return self._queue.popleft()
        if chunk is None:
            raise ValueError("temporal ensembling needs a chunk at every step")
        chunk = np.asarray(chunk)
        self._queue.append(deque(chunk[: self.horizon]))           # maxlen drops the oldest chunk
        now = np.stack([q.popleft() for q in self._queue])          # what every live chunk predicted for this step
        w = np.exp(self.k * np.arange(now.shape[0]))
        return (now * (w / w.sum())[:, None]).sum(axis=0)


class DeviceActionQueue:
    """:class:`ActionQueue` + ``convert_abs_action`` + :func:`to_robot_action` for ``n_episodes`` episodes on the GPU
    (``otter_action_step``): the predicted delta chunks go in as they come out of the policy, 7-d robot actions come out."""

    def __init__(self, n_episodes: int, action_exec_horizon: int, device, receding_horizon_control: bool = False, k: float = 0.05,
                 action_dim: int = 10):
        import torch
        if action_exec_horizon < 1:
            raise ValueError("action_exec_horizon must be at least 1")
        self.n, self.horizon, self.receding, self.k = int(n_episodes), int(action_exec_horizon), bool(receding_horizon_control), float(k)
        self.ring = torch.zeros(self.n, self.horizon, self.horizon, action_dim, dtype=torch.float64, device=device)
        self.counts = torch.zeros(self.n, dtype=torch.int32, device=device)

    def clear(self, episode_ids=None) -> None:
        if episode_ids is None:
            self.counts.zero_()
        else:
            import torch
            self.counts.index_fill_(0, torch.as_tensor(list(episode_ids), dtype=torch.long, device=self.counts.device), 0)

    def step(self, delta_chunks, cur_pose, want_abs: bool = False):
        """delta_chunks fp32 (n, H, 10) on the device, cur_pose fp64 (n, 10) -> (robot (n, 7), executed (n, 10), abs chunks or None)."""
        from .torch_ops import kernels as ops
        return ops.action_step(delta_chunks.contiguous(), cur_pose, self.ring, self.counts, self.horizon, self.receding, self.k, want_abs)


def poses_to_proprio(cartesian: np.ndarray, gripper: np.ndarray) -> np.ndarray:
    """``OtterInference._proprocess_proprio`` (``otter_interface.py:60-84``) for a batch: (n, 6) xyz + Euler XYZ and (n,) or
    (n, 1) gripper -> (n, 10) float64 = xyz, first two rows of the rotation matrix, gripper clipped to [0, 1]."""
    from scipy.spatial.transform import Rotation
    cartesian = np.asarray(cartesian, dtype=np.float64).reshape(-1, 6)
    rot = Rotation.from_euler("XYZ", cartesian[:, 3:]).as_matrix()[:, :2, :].reshape(-1, 6)
    grip = np.clip(np.asarray(gripper, dtype=np.float64).reshape(-1, 1), 0, 1)
    return np.concatenate([cartesian[:, :3], rot, grip], axis=-1)


class BatchedRolloutPolicy:
    """Many robots / simulated episodes behind one model: frames, poses and grippers of ``n`` episodes in, ``n`` 7-d robot
    actions out per control step.  The encoder, the policy with its GPU-resident history, the absolute-action decode, the
    action queue and the 10-d -> 7-d conversion all run on the device; per step the host uploads frames + poses and reads
    back (n, 7) doubles.  The packed weights follow the model: every call asks it for its current device copy (as
    ``OTTER.forward_inference`` does), and when that changed -- an optimizer step, ``load_state_dict``,
    ``invalidate_device_weights()`` -- the running episodes' windows move over to the new weights."""

    def __init__(self, model, n_episodes: int, action_exec_horizon: int, receding_horizon_control: bool = False):
        from .engine import BatchedRollout
        if action_exec_horizon > model.action_horizon:
            raise ValueError(f"action_exec_horizon must be at most the predicted horizon ({model.action_horizon})")
        self.model = model
        self.dev = model.device_model()
        self.rollout = BatchedRollout(self.dev, n_episodes)
        self.queue = DeviceActionQueue(n_episodes, action_exec_horizon, self.dev.device, receding_horizon_control, action_dim=model.action_dim)
        self.n = int(n_episodes)

    def start_episodes(self, episode_ids=None) -> None:
        self.rollout.reset(episode_ids)
        self.queue.clear(episode_ids)

    def refresh(self) -> None:
        """Pick up the model's current packed weights (a no-op while they are unchanged)."""
        from .engine import BatchedRollout
        dev = self.model.device_model()
        if dev is not self.dev:
            fresh = BatchedRollout(dev, self.n)
            fresh.adopt(self.rollout)              # policy inputs of the running episodes are kept, K/V rows are recomputed
            self.dev, self.rollout = dev, fresh

    def __call__(self, frames, token_ids, cartesian: np.ndarray, gripper: np.ndarray):
        """frames uint8 (n*cams, 3, R, R) on the device (camera order = sorted camera keys; :func:`vision_tf.crop_resize_device`
        makes them from raw camera frames), token_ids int32 (n, 77) on the host, cartesian (n, 6), gripper (n,) ->
        robot actions fp64 (n, 7) on the device."""
        import torch
        self.refresh()
        pose = torch.from_numpy(poses_to_proprio(cartesian, gripper)).to(self.dev.device)
        text = self.dev.clip.text_tokens_cached(token_ids)
        chunks = self.rollout.step(frames, text, pose.float())
        return self.queue.step(chunks, pose)[0]


def to_robot_action(action_10d: np.ndarray) -> np.ndarray:
    """[xyz, rot6d, gripper] -> [xyz, euler XYZ, gripper > 0.5] (``droid_inference.py:198-200``)."""
    out = action_10d_to_7d(np.asarray(action_10d, dtype=np.float64))
    out[-1] = out[-1] > 0.5
    return out


class RolloutPolicy:
    """``PolicyWrapper`` of the robot script around an :class:`otter_amd.otter_interface.OtterInference`."""

    def __init__(self, policy, action_exec_horizon: int, receding_horizon_control: bool = False, language_prompt: Optional[str] = None):
        horizon = policy.model.action_horizon
        if action_exec_horizon > horizon:
            raise ValueError(f"action_exec_horizon must be at most the predicted horizon ({horizon})")
        self.policy = policy
        self.queue = ActionQueue(action_exec_horizon, receding_horizon_control)
        self.language_prompt = language_prompt
        self.timestep = 0

    def set_language_prompt(self, language_prompt: str) -> None:
        if not isinstance(language_prompt, str):
            raise TypeError("language prompt must be a string")
        self.language_prompt = language_prompt

    def start_episode(self) -> None:
        self.timestep = 0
        self.queue.clear()
        self.policy.reset()

    def __call__(self, images, proprio: np.ndarray, gripper: np.ndarray) -> np.ndarray:
        """One control step: PIL images per camera key, 6-d cartesian pose, 1-d gripper -> 7-d robot action.
        (Upstream calls the model at every step in both modes; so does this.)"""
        self.timestep += 1
        chunk = self.policy(images=images, text=self.language_prompt, proprio=proprio, gripper=gripper)
        return to_robot_action(self.queue.next_action(chunk))
