"""Import-compatibility shims for names the reference's OTHER packages import from
depth_pose_prediction.networks.layers (slam/utils.py:10 imports BackprojectDepth for point-cloud
export, which is evaluation / visualisation and outside the hot path).

On the hot path these three reference modules (networks/layers.py:51-137) are replaced by the fused
HIP kernels clslam_warp_fwd / clslam_photo_map; the classes here keep the reference's constructor
and call signatures for off-path callers and evaluate with stock torch ops (they are never used by
DepthPosePrediction.predict()/adapt()).
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn


class BackprojectDepth(nn.Module):
    """depth (B,1,H,W), inv_K (B,4,4) -> homogeneous camera points (B,4,H*W)."""

    def __init__(self, batch_size: int, height: int, width: int) -> None:
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width
        xs, ys = np.meshgrid(range(width), range(height), indexing='xy')
        pix = np.stack([xs.reshape(-1), ys.reshape(-1), np.ones(height * width)], 0).astype(np.float32)
        self.register_buffer('pix_coords', torch.from_numpy(pix).unsqueeze(0).repeat(batch_size, 1, 1), persistent=False)
        self.register_buffer('ones', torch.ones(batch_size, 1, height * width), persistent=False)

    def forward(self, depth: Tensor, inv_K: Tensor) -> Tensor:
        cam = torch.matmul(inv_K[:, :3, :3], self.pix_coords)
        cam = depth.view(self.batch_size, 1, -1) * cam
        return torch.cat([cam, self.ones], 1)


class Project3D(nn.Module):
    def __init__(self, batch_size: int, height: int, width: int, eps: float = 1e-7) -> None:
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points: Tensor, K: Tensor, T: Tensor) -> Tensor:
        P = torch.matmul(K, T)[:, :3, :]
        cam = torch.matmul(P, points)
        pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + self.eps)
        pix = pix.view(self.batch_size, 2, self.height, self.width).permute(0, 2, 3, 1)
        pix = torch.stack([pix[..., 0] / (self.width - 1), pix[..., 1] / (self.height - 1)], -1)
        return (pix - 0.5) * 2


class SSIM(nn.Module):
    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        C1, C2 = 0.01**2, 0.03**2
        x = F.pad(x, (1, 1, 1, 1), mode='reflect')
        y = F.pad(y, (1, 1, 1, 1), mode='reflect')
        mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
        sx = F.avg_pool2d(x**2, 3, 1) - mu_x**2
        sy = F.avg_pool2d(y**2, 3, 1) - mu_y**2
        sxy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
        n = (2 * mu_x * mu_y + C1) * (2 * sxy + C2)
        d = (mu_x**2 + mu_y**2 + C1) * (sx + sy + C2)
        return torch.clamp((1 - n / d) / 2, 0, 1)
