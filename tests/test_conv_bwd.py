"""Backward conv pieces (dgrad = transpose + conv2d + fold, wgrad, bias colsum) vs torch autograd."""
import pytest
import torch
import torch.nn.functional as F

from clslam_hip import ops
from emu_util import BACKENDS, use_backend
from helpers import rel_err

# B, H(out res), W, Ca, Cb, Cout, k, reflect, ups, act
CASES = [
    (2, 8, 12, 32, 0, 16, 3, True, False, 2),     # upconv_0_0-like
    (1, 8, 16, 16, 0, 16, 3, True, True, 2),      # upconv_0_1-like (upsampled input)
    (1, 8, 8, 32, 64, 32, 3, True, True, 2),      # upconv_1_1-like (upsample + skip concat)
    (2, 4, 6, 64, 64, 64, 3, True, True, 2),      # upconv_2_1-like, 64-tiles
    (2, 6, 10, 64, 0, 64, 3, False, False, 1),    # pose_0-like (zero pad, relu)
    (2, 6, 10, 64, 0, 64, 1, False, False, 1),    # squeeze-like 1x1
    (2, 10, 36, 16, 0, 16, 3, True, True, 2),     # wgrad_patch 16x16, ragged tiles
    (1, 16, 32, 32, 0, 16, 3, True, False, 2),    # wgrad_patch 16x32
    (1, 8, 48, 32, 64, 32, 3, True, True, 2),     # wgrad_patch 32x32 with concat
    (2, 9, 40, 64, 0, 64, 3, False, False, 1),    # wgrad_patch 32x32, zero pad
    (2, 12, 40, 32, 32, 32, 3, True, True, 2),    # wgrad_patch deep-stage tile 4x40 (upconv_4_1-like: upsample + concat, reflect)
    (1, 12, 40, 64, 0, 32, 3, True, False, 2),    # ... 4x40, one image (upconv_3_0-like)
    (3, 6, 20, 64, 0, 32, 3, True, False, 2),     # wgrad_patch deep-stage tile 6x20 (upconv_4_0-like)
    (4, 6, 20, 32, 0, 64, 3, False, False, 1),    # ... 6x20, zero pad (pose_0-like)
]


def _forward(xa, xb, w_ohwi, bias, k, reflect, ups, act):
    x = xa.permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    if xb is not None:
        x = torch.cat([x, xb.permute(0, 3, 1, 2)], 1)
    Cout = w_ohwi.shape[0]
    w = w_ohwi.reshape(Cout, k, k, -1).permute(0, 3, 1, 2)
    pad = k // 2
    if pad:
        x = F.pad(x, (pad,) * 4, mode='reflect' if reflect else 'constant')
    z = F.conv2d(x, w, bias)
    y = F.relu(z) if act == 1 else F.elu(z)
    return z, y


@pytest.fixture(autouse=True)
def _deep_tiles(monkeypatch):
    monkeypatch.setenv('CLSLAM_WGRAD_DEEP_TILES', '1')      # the opt-in deep-stage tiles of wgrad_patch.hip


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('case', CASES)
def test_conv_backward_matches_autograd(case, backend):
    B, H, W, Ca, Cb, Cout, k, reflect, ups, act = case
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(1234 + hash(case) % 1000)
    Ha, Wa = (H // 2, W // 2) if ups else (H, W)
    # xa is itself the output of an activation (so that dxa gets multiplied by act'(xa))
    xa_pre = torch.randn(B, Ha, Wa, Ca, generator=g)
    xa_pre.requires_grad_(True)
    xa = F.elu(xa_pre) if act == 2 else F.relu(xa_pre)
    xb = torch.randn(B, H, W, Cb, generator=g) if Cb else None
    w = (torch.randn(Cout, k * k, Ca + Cb, generator=g) / (k * (Ca + Cb) ** 0.5)).requires_grad_(True)
    bias = (torch.randn(Cout, generator=g) * 0.1).requires_grad_(True)
    z, y = _forward(xa, xb, w, bias, k, reflect, ups, act)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    # dz = dy * act'(y)  (NHWC)
    with torch.no_grad():
        yn = y.permute(0, 2, 3, 1).contiguous()
        dyn = dy.permute(0, 2, 3, 1).contiguous()
        actg = (yn > 0).float() if act == 1 else torch.where(yn > 0, torch.ones_like(yn), yn + 1)
        dz = (dyn * actg).contiguous()
    t = lambda v: None if v is None else v.detach().contiguous().to(dev)
    dz_d, xa_d, xb_d, w_d = t(dz), t(xa), t(xb), t(w)

    # ---- wgrad -------------------------------------------------------------------------------
    desc = ops.conv_desc(xa_d, (B, H, W, Cout), src_b=xb_d, ksize=k, pad_mode=1 if reflect else 0, upsample_a=ups)
    for target in (1, 64):
        splits = ops.wgrad_splits(desc, target)
        n = w.numel()
        partial = torch.full((splits * n,), float('nan'), device=dev)
        ops.conv_wgrad(desc, dz_d, partial, splits)
        dw = torch.empty(n, device=dev)
        ops.reduce_partials(partial, dw, n, splits)
        assert rel_err(dw.cpu().view_as(w), w.grad) < 2e-5, (target, splits)
    if k == 3 and Ca % 32 == 0 and Cb % 32 == 0 and Cout % 32 == 0 and ((W, H % 4) == (40, 0) or (H, W) == (6, 20)):
        assert ops.wgrad_patch_supported(desc)          # the deep-stage tiles are on in this file
    if ops.wgrad_patch_supported(desc):
        for target in (1, 16):
            splits = ops.wgrad_patch_splits(desc, target)
            partial = torch.full((splits * w.numel(),), float('nan'), device=dev)
            ops.conv_wgrad_patch(desc, dz_d, partial, splits)
            dw = torch.empty(w.numel(), device=dev)
            ops.reduce_partials(partial, dw, w.numel(), splits)
            assert rel_err(dw.cpu().view_as(w), w.grad) < 2e-5, ('patch', target, splits)
    # ---- bias grad ---------------------------------------------------------------------------
    rows = B * H * W
    nb = ops.colsum_blocks(rows)
    part = torch.empty(nb * Cout, device=dev)
    ops.colsum(dz_d, part, rows, Cout)
    db = torch.empty(Cout, device=dev)
    ops.reduce_partials(part, db, Cout, nb)
    assert rel_err(db.cpu(), bias.grad) < 2e-5
    # ---- dgrad w.r.t. the pre-activation of xa -----------------------------------------------
    taps = k * k
    wt = torch.empty(Ca * taps * Cout, device=dev)
    ops.weight_transpose(w_d, wt.view(Ca, taps, Cout), ch_in_sel=Ca)
    if k == 3 and reflect:
        dxp = torch.full((B, H + 2, W + 2, Ca), float('nan'), device=dev)
        ops.conv2d(dz_d, wt.view(Ca, taps, Cout), dxp, ksize=3, pad=2)
        for pcfg in ((10, 13) if Ca % 64 == 0 else (11,) if Ca % 32 == 0 else (12,)):
            dxp2 = torch.full_like(dxp, float('nan'))
            ops.conv2d(dz_d, wt.view(Ca, taps, Cout), dxp2, ksize=3, pad=2, config=pcfg)
            assert rel_err(dxp2.cpu(), dxp.cpu()) < 1e-5, pcfg
        dpre = torch.empty(B, Ha, Wa, Ca, device=dev)
        nbf = ops.fold_blocks(B, H, W, Ca, ups)
        bpart = torch.full((nbf * Ca,), float('nan'), device=dev)
        ops.fold_act_grad(dxp, xa_d, dpre, h=H, w=W, ch=Ca, border=1, pool=ups, act=act, bias_partial=bpart)
        bsum = torch.empty(Ca, device=dev)
        ops.reduce_partials(bpart, bsum, Ca, nbf)       # fused column sums == bias gradient of the producer conv
        assert rel_err(bsum.cpu(), dpre.cpu().sum((0, 1, 2))) < 1e-5
    else:
        dpre = torch.full((B, H, W, Ca), float('nan'), device=dev)
        ops.conv2d(dz_d, wt.view(Ca, taps, Cout), dpre, ksize=k, pad=k // 2, actgrad_src=xa_d, actgrad_kind=act)
    assert rel_err(dpre.cpu(), xa_pre.grad) < 2e-5


@pytest.mark.parametrize('backend', BACKENDS)
def test_reduce_multi_all_split_regimes(backend):
    """clslam_reduce_multi: one launch sums every layer's split partials.  Items cover the three lane
    layouts (<= 24 splits: one thread per float4 output; <= 96: 4 split lanes; more: 16 split lanes),
    the tiny-n bias case and the scalar (n % 4 != 0) path."""
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(3)
    shapes = [(4096, 5), (2052, 24), (1024, 40), (256, 96), (64, 300), (16, 1024), (18, 7), (12 * 256 + 1, 3)]
    total = sum(n for n, _ in shapes)
    out = torch.full((total,), float('nan'), device=dev)
    items, refs, off = [], [], 0
    for n, splits in shapes:
        part = torch.randn(splits, n, generator=g)
        items.append((part.to(dev).reshape(-1).contiguous(), out[off:off + n], n, splits))
        refs.append(part.double().sum(0))
        off += n
    table = ops.make_reduce_table(items, dev)
    ops.reduce_multi(table, len(items), out)
    got = out.cpu().double()
    ref = torch.cat(refs)
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) < 2e-4 * float(ref.abs().max())
    again = torch.empty_like(out)
    items2 = [(p, again[o.storage_offset():o.storage_offset() + n], n, s) for (p, o, n, s) in items]
    ops.reduce_multi(ops.make_reduce_table(items2, dev), len(items2), again)
    assert torch.equal(again, out)          # fixed summation order: bitwise reproducible


@pytest.mark.parametrize('backend', BACKENDS)
def test_weight_transpose_multi_equals_single_launches(backend):
    """clslam_weight_transpose_multi: every item of the table is bitwise what clslam_weight_transpose produces for it
    (including a channel-selected item, the dgrad weights of a skip-concat conv)."""
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(5)
    shapes = [(16, 9, 16, None), (32, 9, 96, 32), (256, 9, 256, None), (64, 1, 128, None), (16, 9, 32, 16)]
    items, refs = [], []
    for cout, taps, cin, sel in shapes:
        w = torch.randn(cout, taps, cin, generator=g).to(dev)
        s = cin if sel is None else sel
        wt = torch.full((s, taps, cout), float('nan'), device=dev)
        ref = torch.empty(s, taps, cout, device=dev)
        ops.weight_transpose(w, ref, ch_in_sel=sel)
        expect = w[:, :, :s].flip(1).permute(2, 1, 0).contiguous()
        assert torch.equal(ref.cpu(), expect.cpu())
        items.append((w, wt, sel)); refs.append(ref)
    table = ops.transpose_table(items)
    ops.weight_transpose_multi(table, items[0][0])
    for (w, wt, sel), ref in zip(items, refs):
        assert torch.equal(wt.cpu(), ref.cpu())
