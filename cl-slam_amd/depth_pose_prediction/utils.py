"""Host-side helpers with the reference's names (depth_pose_prediction/utils.py:34-142).  The hot
path computes these inside clslam_pose_to_proj / clslam_warp_fwd; the functions below serve callers
that want the same conversion on a handful of host-side values (e.g. evaluation scripts)."""
from typing import Optional, Union

import numpy as np
import torch
from torch import Tensor


def rot_from_axisangle(axis_angle: Tensor) -> Tensor:
    """(B,1,3) -> (B,4,4) Rodrigues rotation, axis = v / (|v| + 1e-7)."""
    angle = torch.norm(axis_angle, 2, 2, True)
    axis = axis_angle / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = (axis[..., k].unsqueeze(1) for k in range(3))
    rot = torch.zeros((axis_angle.shape[0], 4, 4), device=axis_angle.device, dtype=axis_angle.dtype)
    rows = ((x * x * C + ca, x * y * C - z * sa, z * x * C + y * sa),
            (x * y * C + z * sa, y * y * C + ca, y * z * C - x * sa),
            (z * x * C - y * sa, y * z * C + x * sa, z * z * C + ca))
    for i in range(3):
        for j in range(3):
            rot[:, i, j] = torch.squeeze(rows[i][j])
    rot[:, 3, 3] = 1
    return rot


def get_translation_matrix(translation_vector: Tensor) -> Tensor:
    T = torch.eye(4, device=translation_vector.device, dtype=translation_vector.dtype).repeat(translation_vector.shape[0], 1, 1)
    T[:, :3, 3] = translation_vector.contiguous().view(-1, 3)
    return T


def transformation_from_parameters(axis_angle: Tensor, translation: Tensor, invert: bool = False) -> Tensor:
    R = rot_from_axisangle(axis_angle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


def disp_to_depth(disp: Union[Tensor, np.ndarray], min_depth: Optional[float] = None,
                  max_depth: Optional[float] = None) -> Union[Tensor, np.ndarray]:
    if min_depth is None and max_depth is None:
        return 1 / disp
    if max_depth is None:
        return min_depth / disp
    if min_depth is None:
        raise ValueError('min_depth is None')
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    return 1 / (min_disp + (max_disp - min_disp) * disp)
