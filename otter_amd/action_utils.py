"""Host-side action / rotation helpers of the rollout interface (numpy + scipy, microseconds per call).

Restates ``otter/dataset/utils.py``: ``rot_mat_to_rot_6d`` :9-17, ``rot_6d_to_rot_mat`` :19-34, ``gram_schmidt``
:100-119, ``action_10d_to_7d`` :145-151, ``convert_abs_action`` :236-264.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def gram_schmidt(vectors: np.ndarray) -> np.ndarray:
    """Row-wise Gram-Schmidt on (batch, N, D); like the original it updates the rows of its argument in place."""
    if vectors.ndim == 2:
        vectors = vectors[None]
    basis = np.zeros_like(vectors)
    basis[:, 0] = vectors[:, 0] / np.linalg.norm(vectors[:, 0], axis=-1, keepdims=True)
    for i in range(1, vectors.shape[1]):
        v = vectors[:, i]
        for j in range(i):
            v -= np.sum(v * basis[:, j], axis=-1, keepdims=True) * basis[:, j]
        basis[:, i] = v / np.linalg.norm(v, axis=-1, keepdims=True)
    return basis


def rot_mat_to_rot_6d(rot_mat: np.ndarray) -> np.ndarray:
    return rot_mat[:, :2, :].reshape(-1, 6)


def rot_6d_to_rot_mat(rot_6d: np.ndarray) -> np.ndarray:
    """(N, 6) -> (N, 3, 3): first two rows as given (orthonormalised when they are not orthogonal), third = cross."""
    r = rot_6d.reshape(-1, 2, 3)
    if not np.allclose(np.sum(r[:, 0] * r[:, 1], axis=-1), 0):
        r = gram_schmidt(r)
    out = np.zeros((r.shape[0], 3, 3))
    out[:, :2, :] = r
    out[:, 2, :] = np.cross(r[:, 0], r[:, 1])
    return out


def rot_6d_to_euler(rot_6d: np.ndarray, fmt: str = "XYZ") -> np.ndarray:
    return Rotation.from_matrix(rot_6d_to_rot_mat(rot_6d)).as_euler(fmt, degrees=False)


def action_10d_to_7d(action: np.ndarray) -> np.ndarray:
    """[xyz, rot6d, gripper] -> [xyz, euler XYZ, gripper]."""
    return np.concatenate([action[:3], rot_6d_to_euler(action[3:-1]).squeeze(), action[-1:]], axis=-1)


def convert_abs_action(action: np.ndarray, proprio: np.ndarray) -> np.ndarray:
    """Delta actions (S, T, A) relative to the current pose ``proprio`` (S, 1|T, A) or (S, A) -> absolute actions.

    translation = R_cur . dt + t_cur ; rotation = R_cur . R_delta (scipy ``Rotation`` orthonormalises both);
    everything after the 9 pose entries (gripper, optional end-of-sequence flag) passes through.
    """
    steps = action.shape[1]
    d_trans = action[:, :, :3].reshape(-1, 3)
    d_rot = Rotation.from_matrix(rot_6d_to_rot_mat(action[:, :, 3:9].reshape(-1, 6)))
    if proprio.ndim == 2:
        proprio = proprio[:, None, :]
    cur = np.repeat(proprio[:, 0:1], steps, 1)
    cur_trans = cur[:, :, :3].reshape(-1, 3)
    cur_rot = Rotation.from_matrix(rot_6d_to_rot_mat(cur[:, :, 3:9].reshape(-1, 6)))
    trans = np.einsum("ijk,ik->ij", cur_rot.as_matrix(), d_trans) + cur_trans
    rot = rot_mat_to_rot_6d((cur_rot * d_rot).as_matrix()).reshape(-1, steps, 6)
    trans = trans.reshape(-1, steps, 3)
    tail = action[:, :, -1:] if action.shape[-1] == proprio.shape[-1] else action[:, :, -2:]
    return np.concatenate([trans, rot, tail], axis=-1)
