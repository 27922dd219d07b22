"""Oracle geometry + losses (torch-CPU fp32, differentiable through autograd).

Citations are to the reference repository (paths relative to its root); ``dpp.py`` =
``depth_pose_prediction/depth_pose_prediction.py``.
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


# ----------------------------------------------------------------------------- pose -> 4x4
def rot_from_axisangle(axis_angle: Tensor) -> Tensor:
    """depth_pose_prediction/utils.py:74-117.  axis_angle (B,1,3) -> (B,4,4)."""
    angle = torch.norm(axis_angle, 2, 2, True)
    axis = axis_angle / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x = axis[..., 0].unsqueeze(1)
    y = axis[..., 1].unsqueeze(1)
    z = axis[..., 2].unsqueeze(1)
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    rot = torch.zeros((axis_angle.shape[0], 4, 4), dtype=axis_angle.dtype)
    rot[:, 0, 0] = torch.squeeze(x * xC + ca)
    rot[:, 0, 1] = torch.squeeze(xyC - zs)
    rot[:, 0, 2] = torch.squeeze(zxC + ys)
    rot[:, 1, 0] = torch.squeeze(xyC + zs)
    rot[:, 1, 1] = torch.squeeze(y * yC + ca)
    rot[:, 1, 2] = torch.squeeze(yzC - xs)
    rot[:, 2, 0] = torch.squeeze(zxC - ys)
    rot[:, 2, 1] = torch.squeeze(yzC + xs)
    rot[:, 2, 2] = torch.squeeze(z * zC + ca)
    rot[:, 3, 3] = 1
    return rot


def get_translation_matrix(t: Tensor) -> Tensor:
    """depth_pose_prediction/utils.py:58-71."""
    T = torch.zeros(t.shape[0], 4, 4, dtype=t.dtype)
    T[:, 0, 0] = 1
    T[:, 1, 1] = 1
    T[:, 2, 2] = 1
    T[:, 3, 3] = 1
    T[:, :3, 3, None] = t.contiguous().view(-1, 3, 1)
    return T


def transformation_from_parameters(axis_angle: Tensor, translation: Tensor,
                                   invert: bool = False) -> Tensor:
    """depth_pose_prediction/utils.py:34-55."""
    R = rot_from_axisangle(axis_angle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


def disp_to_depth(disp: Tensor, min_depth: Optional[float], max_depth: Optional[float]) -> Tensor:
    """depth_pose_prediction/utils.py:120-142."""
    if min_depth is None and max_depth is None:
        return 1 / disp
    if max_depth is None:
        return min_depth / disp
    if min_depth is None:
        raise ValueError('min_depth is None')
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    return 1 / (min_disp + (max_disp - min_disp) * disp)


# ----------------------------------------------------------------------------- view synthesis
def backproject(depth: Tensor, inv_K: Tensor) -> Tensor:
    """networks/layers.py:51-79.  depth (B,1,H,W) -> homogeneous points (B,4,H*W)."""
    B, _, H, W = depth.shape
    # (depth.dtype: fp32 like the reference; the fp64 re-run of tests/helpers.py attributes fp32 rounding)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=depth.dtype),
                            torch.arange(W, dtype=depth.dtype), indexing='ij')
    ones = torch.ones(B, 1, H * W, dtype=depth.dtype)
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1)], 0).unsqueeze(0).repeat(B, 1, 1)
    pix = torch.cat([pix, ones], 1)
    cam = torch.matmul(inv_K[:, :3, :3], pix)
    cam = depth.view(B, 1, -1) * cam
    return torch.cat([cam, ones], 1)


def project(points: Tensor, K: Tensor, T: Tensor, H: int, W: int, eps: float = 1e-7, P_value: Optional[Tensor] = None) -> Tensor:
    """networks/layers.py:82-104.  -> normalised sampling grid (B,H,W,2).
    P_value: test hook -- the VALUE of the projection matrix (K T)[:3] as another implementation rounded it (its fp32
    product differs from this one's by an ulp per entry, which moves every sample of the frame coherently by ~1e-5 px: a
    forward-point difference, tests/helpers.py); the gradient path through K T stays."""
    B = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    if P_value is not None:
        P = P + (P_value.to(P.dtype) - P).detach()
    cam = torch.matmul(P, points)
    pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)
    pix = pix.view(B, 2, H, W).permute(0, 2, 3, 1)
    pix = torch.stack([pix[..., 0] / (W - 1), pix[..., 1] / (H - 1)], -1)
    return (pix - 0.5) * 2


def grid_sample_border(src: Tensor, grid: Tensor, cells=None, record=None) -> Tensor:
    """F.grid_sample(src, grid, mode='bilinear', padding_mode='border', align_corners=True) (dpp.py:1013-1017) written
    out the way ATen's grid sampler computes it: un-normalise, clip to the image with the gradient
    zeroed at / outside the border (clip_coordinates_set_grad), floor, four taps weighted by the opposite corner's
    distance, taps beyond the last row / column dropped.  tests/test_oracle_golden.py holds it to F.grid_sample in value
    and gradient.

    The function is piecewise smooth in the sampling position: which CELL (x0, y0) a sample falls into and whether it is
    CLIPPED are decisions, and two implementations whose positions differ by 1e-5 px take them differently for a few
    pixels.  `cells = (x0, y0, mx, my)` (LongTensor x2, BoolTensor x2, each (B,H,W)) imposes another implementation's
    decisions -- the kernel path's, tests/test_backward_parity.py -- the way `forced_sel` imposes its 4-way-min selection;
    `record` (a list) receives this call's own decisions."""
    B, C, H, W = src.shape
    ix = ((grid[..., 0] + 1) / 2) * (W - 1)
    iy = ((grid[..., 1] + 1) / 2) * (H - 1)
    if cells is None:
        mx = (ix > 0) & (ix < W - 1)
        my = (iy > 0) & (iy < H - 1)
        x0 = torch.floor(ix.detach().clamp(0, W - 1)).long()
        y0 = torch.floor(iy.detach().clamp(0, H - 1)).long()
    else:
        x0, y0, mx, my = cells
    if record is not None:
        record.append((x0, y0, mx, my))
    if cells is None:
        ixc = torch.where(mx, ix, ix.detach().clamp(0, W - 1))      # clipped coordinate; no gradient where clipped
        iyc = torch.where(my, iy, iy.detach().clamp(0, H - 1))
    else:   # an imposed clip puts the sample ON the border pixel the other implementation chose (its x0 is 0 or W-1 there)
        ixc = torch.where(mx, ix, x0.to(ix.dtype))
        iyc = torch.where(my, iy, y0.to(iy.dtype))
    wx1, wy1 = ixc - x0.to(ix.dtype), iyc - y0.to(iy.dtype)
    wx0, wy0 = (x0 + 1).to(ix.dtype) - ixc, (y0 + 1).to(iy.dtype) - iyc
    x1ok, y1ok = (x0 + 1 <= W - 1), (y0 + 1 <= H - 1)
    x1, y1 = (x0 + 1).clamp(max=W - 1), (y0 + 1).clamp(max=H - 1)
    flat = src.reshape(B, C, H * W)

    def tap(yy, xx):
        idx = (yy * W + xx).reshape(B, 1, H_out * W_out).expand(B, C, H_out * W_out)
        return torch.gather(flat, 2, idx).reshape(B, C, H_out, W_out)
    H_out, W_out = grid.shape[1], grid.shape[2]
    zero = torch.zeros((), dtype=src.dtype)
    out = tap(y0, x0) * (wx0 * wy0).unsqueeze(1)
    out = out + tap(y0, x1) * torch.where(x1ok, wx1 * wy0, zero).unsqueeze(1)
    out = out + tap(y1, x0) * torch.where(y1ok, wx0 * wy1, zero).unsqueeze(1)
    out = out + tap(y1, x1) * torch.where(x1ok & y1ok, wx1 * wy1, zero).unsqueeze(1)
    return out


def reconstruct(disp_s: Tensor, T: Dict[int, Tensor], K: Tensor, inv_K: Tensor,
                src: Dict[int, Tensor], H: int, W: int, min_depth, max_depth,
                cells=None, record=None, P_value=None) -> Tuple[Tensor, Dict[int, Tensor]]:
    """dpp.py:986-1017 for one scale: bilinear-upsample disp, disp->depth, backproject,
    project with scale-0 intrinsics, grid_sample the un-augmented scale-0 source frame.
    cells / record: {frame: ...} test hooks of grid_sample_border (the written-out sampler replaces F.grid_sample
    whenever one of them is given)."""
    disp = F.interpolate(disp_s, [H, W], mode='bilinear', align_corners=False)
    depth = disp_to_depth(disp, min_depth, max_depth)
    pts = backproject(depth, inv_K)
    warped = {}
    for f in (-1, 1):
        grid = project(pts, K, T[f], H, W, P_value=None if P_value is None else P_value.get(f))
        if cells is None and record is None:
            warped[f] = F.grid_sample(src[f], grid, padding_mode='border', align_corners=True)
        else:
            rec = None if record is None else record.setdefault(f, [])
            warped[f] = grid_sample_border(src[f], grid, None if cells is None else cells[f], rec)
    return depth, warped


# ----------------------------------------------------------------------------- losses
def ssim(x: Tensor, y: Tensor) -> Tensor:
    """networks/layers.py:107-137."""
    C1, C2 = 0.01**2, 0.03**2
    x = F.pad(x, (1, 1, 1, 1), mode='reflect')
    y = F.pad(y, (1, 1, 1, 1), mode='reflect')
    mu_x = F.avg_pool2d(x, 3, 1)
    mu_y = F.avg_pool2d(y, 3, 1)
    sigma_x = F.avg_pool2d(x**2, 3, 1) - mu_x**2
    sigma_y = F.avg_pool2d(y**2, 3, 1) - mu_y**2
    sigma_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
    d = (mu_x**2 + mu_y**2 + C1) * (sigma_x + sigma_y + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def reprojection_loss(pred: Tensor, target: Tensor, l1_sign: Optional[Tensor] = None) -> Tensor:
    """dpp.py:1178-1192.  l1_sign: test hook -- sign(target - pred) per element as ANOTHER implementation decided it (the
    kink of |.|: two implementations whose synthesised images differ by 1e-7 disagree where target and prediction cross);
    the value is unchanged wherever the signs agree, the gradient follows the imposed sign."""
    l1 = (torch.abs(target - pred) if l1_sign is None else l1_sign * (target - pred)).mean(1, True)
    return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1


def smooth_loss_reference(disp: Tensor, img: Tensor) -> Tensor:
    """dpp.py:1148-1176 with the all-true mask of the adaptation config
    (mask_dynamic=False, :1081-1084), INCLUDING the flattening quirk:
    ``masked_select`` flattens the whole batch, so ``grad_disp_x[i, ...]`` is the single flat
    element i (pixel (0,i) of sample 0), broadcast against the mask and averaged."""
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    gdx = (gdx * torch.exp(-gix)).reshape(-1)
    gdy = (gdy * torch.exp(-giy)).reshape(-1)
    B = disp.shape[0]
    return torch.stack([gdx[i] + gdy[i] for i in range(B)])


def smooth_loss_intended(disp: Tensor, img: Tensor) -> Tensor:
    """The per-sample edge-aware smoothness the reference evidently meant (monodepth2);
    opt-in only, never used for parity."""
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    gdx = gdx * torch.exp(-gix)
    gdy = gdy * torch.exp(-giy)
    return gdx.flatten(1).mean(1) + gdy.flatten(1).mean(1)


def velocity_loss(trans_m1: Tensor, trans_p1: Tensor, dist0: Tensor, dist1: Tensor) -> Tensor:
    """dpp.py:1125-1146: frame 0 pairs translation(0->-1) with relative_distance(0), frame 1
    pairs translation(0->+1) with relative_distance(1); L1 in the promoted dtype
    (relative_distance is float64), accumulated into an fp32 vector, /2."""
    B = trans_m1.shape[0]
    out = torch.zeros(B, dtype=trans_m1.dtype)       # fp32 in the reference
    for pred_t, gt in ((trans_m1, dist0), (trans_p1, dist1)):
        gt_d = torch.abs(gt).reshape(B)
        pred_d = torch.linalg.norm(pred_t, dim=-1).reshape(B)
        # in-place fp32 += fp64 computes in fp64 and rounds back to fp32 (dpp.py:1142)
        out = (out.double() + F.l1_loss(pred_d.double(), gt_d.double(), reduction='none')).to(trans_m1.dtype)
    return out / 2
