"""Stem conv / maxpool / disparity head / pose head kernels vs torch fp32 references."""
import pytest
import torch
import torch.nn.functional as F

from clslam_hip import ops
from emu_util import BACKENDS, use_backend
from helpers import rel_err


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('n_img,B,H,W', [(1, 2, 32, 64), (2, 1, 32, 48), (1, 1, 20, 36)])
def test_stem_and_maxpool(backend, n_img, B, H, W):
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(5)
    imgs = [torch.rand(B, 3, H, W, generator=g) for _ in range(n_img)]
    w = torch.randn(64, 3 * n_img, 7, 7, generator=g) * 0.05
    scale = torch.rand(64, generator=g) + 0.5
    shift = torch.randn(64, generator=g) * 0.1
    x = (torch.cat(imgs, 1) - 0.45) / 0.225
    ref = F.relu(F.conv2d(x, w, stride=2, padding=3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    Ho, Wo = ref.shape[2:]
    out = torch.full((B, Ho, Wo, 64), float('nan'), device=dev)
    ops.stem_conv(imgs[0].to(dev), imgs[1].to(dev) if n_img == 2 else None, ops.stem_pack_weight(w.to(dev)), scale.to(dev),
                  shift.to(dev), out)
    assert rel_err(out.cpu().permute(0, 3, 1, 2), ref) < 2e-5
    refp = F.max_pool2d(ref, 3, 2, 1)
    outp = torch.full((B, refp.shape[2], refp.shape[3], 64), float('nan'), device=dev)
    ops.maxpool3x3s2(out, outp)
    assert rel_err(outp.cpu().permute(0, 3, 1, 2), F.max_pool2d(out.cpu().permute(0, 3, 1, 2), 3, 2, 1)) == 0.0
    assert rel_err(outp.cpu().permute(0, 3, 1, 2), refp) < 2e-5


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('B,H,W,C', [(2, 8, 12, 16), (1, 6, 10, 128), (3, 4, 8, 32), (1, 7, 9, 64),
                                     (2, 20, 70, 16), (1, 9, 40, 32), (2, 10, 35, 64), (1, 11, 19, 128)])  # several tiles, ragged edges
def test_dispconv_fwd_bwd(backend, B, H, W, C):
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, H, W, C, generator=g).requires_grad_(True)
    w = (torch.randn(9, C, generator=g) * 0.1).requires_grad_(True)   # [tap][c]
    bias = torch.randn(1, generator=g).requires_grad_(True)
    wt = w.view(3, 3, C).permute(2, 0, 1).unsqueeze(0).contiguous()   # (1,C,3,3)
    z = F.conv2d(F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1), mode='reflect'), wt, bias)
    disp = torch.sigmoid(z)
    gd = torch.randn(disp.shape, generator=g)
    disp.backward(gd)
    dz = (gd * disp * (1 - disp)).detach().squeeze(1).contiguous()   # (B,H,W)
    xd, wd, bd = x.detach().to(dev), w.detach().to(dev), bias.detach().to(dev)
    out = torch.full((B, H, W), float('nan'), device=dev)
    ops.dispconv_fwd(xd, wd, bd, out)
    assert rel_err(out.cpu(), disp.detach().squeeze(1)) < 1e-5
    # data gradient through the padded domain + fold
    dxp = torch.full((B, H + 2, W + 2, C), float('nan'), device=dev)
    ops.dispconv_bwd_data(dz.to(dev), wd, dxp, C, accumulate=False)
    ops.dispconv_bwd_data(dz.to(dev), wd, dxp, C, accumulate=True)
    dx = torch.empty(B, H, W, C, device=dev)
    ops.fold_act_grad(dxp, None, dx, h=H, w=W, ch=C, border=1, pool=False, act=0)
    assert rel_err(dx.cpu(), 2 * x.grad) < 2e-5
    # the same gradient fused into the fold (what the engine uses): alone, and on top of an upstream dxp
    dx2 = torch.full((B, H, W, C), float('nan'), device=dev)
    ops.fold_act_grad(None, None, dx2, h=H, w=W, ch=C, border=1, pool=False, act=0, disp_dz=dz.to(dev), disp_w=wd)
    assert rel_err(dx2.cpu(), x.grad) < 2e-5
    dx3 = torch.empty(B, H, W, C, device=dev)
    ops.fold_act_grad(dxp, None, dx3, h=H, w=W, ch=C, border=1, pool=False, act=0, disp_dz=dz.to(dev), disp_w=wd)
    assert rel_err(dx3.cpu(), 3 * x.grad) < 2e-5
    nb = ops.dispconv_wgrad_blocks(B * H * W)
    n = 9 * C + 1
    part = torch.full((nb * n,), float('nan'), device=dev)
    ops.dispconv_wgrad(dz.to(dev), xd, part)
    gw = torch.empty(n, device=dev)
    ops.reduce_partials(part, gw, n, nb)
    assert rel_err(gw[:9 * C].cpu().view(9, C), w.grad) < 2e-5
    assert rel_err(gw[9 * C:].cpu(), bias.grad) < 2e-5


@pytest.mark.parametrize('backend', BACKENDS)
def test_pose_head(backend):
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(9)
    N, H, W = 4, 2, 4
    pre = torch.randn(N, H, W, 256, generator=g).requires_grad_(True)
    x = F.relu(pre)
    w2 = (torch.randn(12, 256, generator=g) * 0.1).requires_grad_(True)
    b2 = torch.randn(12, generator=g).requires_grad_(True)
    out = F.conv2d(x.permute(0, 3, 1, 2), w2.view(12, 256, 1, 1), b2).mean(3).mean(2)
    pose = 0.01 * out
    gp = torch.randn(N, 12, generator=g)
    gp[:, 6:] = 0
    pose.backward(gp)
    xd = x.detach().to(dev)
    mean = torch.empty(N, 256, device=dev)
    po = torch.empty(N, 12, device=dev)
    ops.pose_head_fwd(xd, w2.detach().to(dev), b2.detach().to(dev), mean, po)
    assert rel_err(po.cpu(), pose.detach()) < 1e-5
    dz1 = torch.empty(N, H, W, 256, device=dev)
    dw2 = torch.empty(12, 256, device=dev)
    db2 = torch.empty(12, device=dev)
    ops.pose_head_bwd(gp.to(dev), xd, w2.detach().to(dev), mean, dz1, dw2, db2)
    assert rel_err(dz1.cpu(), pre.grad) < 1e-5
    assert rel_err(dw2.cpu(), w2.grad) < 1e-5
    assert rel_err(db2.cpu(), b2.grad) < 1e-5
