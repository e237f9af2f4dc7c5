"""Action post-processing of the robot rollout loop (``script/droid_inference.py:97-200``): which element of the
predicted chunks is executed at each control step.

Two modes, as upstream: receding-horizon control (execute the first ``action_exec_horizon`` actions of a chunk, then
predict again) and temporal ensembling (every step predicts a chunk; the action executed now is the
exponentially weighted mean of what the last ``action_exec_horizon`` chunks predicted for this step, weights
``exp(k * i)`` with i = 0 for the oldest chunk, k = 0.05).  Host-side numpy: a few hundred bytes per step.
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
