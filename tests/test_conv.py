"""clslam_conv2d vs torch fp32 reference convs (zero/reflect pad, stride, upsample+concat,
BN/bias/residual/activation epilogue), every tile configuration."""
import pytest
import torch
import torch.nn.functional as F

from clslam_hip import ops
from emu_util import BACKENDS, use_backend
from helpers import rel_err


def _ref_conv(xa, w_ohwi, *, xb=None, scale=None, shift=None, residual=None, ksize=3, stride=1, pad=1,
              pad_mode=0, ups=False, act=0):
    x = xa.permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    if xb is not None:
        x = torch.cat([x, xb.permute(0, 3, 1, 2)], 1)
    Cout = w_ohwi.shape[0]
    w = w_ohwi.reshape(Cout, ksize, ksize, -1).permute(0, 3, 1, 2)
    if pad > 0:
        x = F.pad(x, (pad,) * 4, mode='reflect' if pad_mode == 1 else 'constant')
    y = F.conv2d(x, w, stride=stride)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual.permute(0, 3, 1, 2)
    y = F.relu(y) if act == 1 else F.elu(y) if act == 2 else y
    return y.permute(0, 2, 3, 1).contiguous()


CASES = [
    # B, H, W, Ca, Cb, Cout, k, stride, pad_mode, ups, act, resid, config
    (2, 8, 12, 32, 0, 64, 3, 1, 0, False, 1, True, 1),
    (1, 9, 7, 64, 0, 64, 3, 2, 0, False, 1, False, 2),
    (2, 6, 10, 32, 0, 64, 1, 2, 0, False, 0, False, 1),
    (1, 8, 8, 32, 32, 32, 3, 1, 1, True, 2, False, 3),
    (1, 8, 16, 16, 0, 16, 3, 1, 1, True, 2, False, 4),
    (1, 8, 16, 32, 0, 16, 3, 1, 1, False, 2, False, 6),
    (1, 6, 10, 32, 64, 32, 3, 1, 1, True, 2, False, 3),
    (3, 12, 20, 64, 0, 64, 3, 1, 0, False, 1, True, 0),
    (1, 4, 6, 16, 0, 32, 3, 1, 0, False, 0, False, 5),
    (2, 6, 20, 64, 0, 128, 3, 1, 0, False, 1, False, -1),
    # config -1 (the library picks) for every conv family of the networks: 1x1 stride 2 / stride 1, 3x3 stride 2, wide 3x3
    (2, 12, 20, 64, 0, 128, 1, 2, 0, False, 0, False, -1),
    (2, 6, 10, 64, 0, 32, 1, 1, 0, False, 1, False, -1),
    (2, 12, 40, 32, 0, 64, 3, 2, 0, False, 1, False, -1),
    (1, 16, 48, 32, 16, 16, 3, 1, 1, True, 2, False, -1),
    # LDS-patch kernel (configs 10-13): ragged tiles, reflect/zero, upsample + concat, residual
    (2, 12, 20, 32, 0, 64, 3, 1, 0, False, 1, True, 10),
    (1, 10, 36, 64, 64, 64, 3, 1, 1, True, 2, False, 10),
    (1, 8, 16, 32, 64, 32, 3, 1, 1, True, 2, False, 11),
    (2, 9, 18, 16, 0, 16, 3, 1, 1, False, 2, False, 12),
    (1, 12, 40, 16, 0, 16, 3, 1, 1, True, 2, False, 12),
    (3, 6, 20, 48, 0, 128, 3, 1, 0, False, 1, True, 13),
    (1, 9, 18, 32, 0, 16, 3, 1, 1, False, 2, False, 14),
    (1, 18, 20, 16, 16, 16, 3, 1, 1, True, 2, False, 15),
    (2, 9, 18, 16, 0, 32, 3, 1, 0, False, 1, True, 16),
    (2, 6, 20, 32, 0, 48, 3, 1, 0, False, 1, False, 17),
    (3, 6, 20, 32, 0, 48, 3, 1, 0, False, 1, True, 18),
    (2, 12, 40, 16, 16, 32, 3, 1, 1, True, 2, False, 18),
    (2, 2, 4, 32, 0, 16, 3, 1, 1, False, 2, False, 18),
    (2, 12, 40, 32, 0, 32, 3, 1, 0, False, 1, False, 19),
    (2, 12, 40, 48, 0, 64, 3, 1, 0, False, 1, True, 26),     # 4x8 px x 32 ch tiles, width 40
    (1, 10, 24, 32, 32, 32, 3, 1, 1, True, 2, False, 26),    # ragged rows, upsample + concat, reflect
    (1, 14, 44, 16, 0, 32, 3, 1, 0, False, 0, False, 26),    # dgrad-like padded domain (12x40 + 2)
    (1, 5, 22, 16, 0, 16, 3, 1, 0, False, 0, False, 19),
    (1, 10, 36, 32, 0, 32, 3, 1, 1, False, 2, False, 20),
    (2, 6, 20, 32, 0, 48, 3, 1, 0, False, 1, False, 21),
    (2, 6, 20, 32, 0, 48, 3, 1, 0, False, 1, False, 22),
    (2, 12, 40, 32, 0, 64, 3, 2, 0, False, 1, False, 23),
    (1, 9, 21, 16, 0, 32, 3, 2, 0, False, 1, False, 23),
]


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('case', CASES)
def test_conv2d_matches_torch(case, backend):
    B, H, W, Ca, Cb, Cout, k, stride, pad_mode, ups, act, resid, config = case
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    Ha, Wa = (H // 2, W // 2) if ups else (H, W)
    xa = torch.randn(B, Ha, Wa, Ca, generator=g)
    xb = torch.randn(B, H, W, Cb, generator=g) if Cb else None
    w = torch.randn(Cout, k * k, Ca + Cb, generator=g) / (k * (Ca + Cb) ** 0.5)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Wo, Cout, generator=g) if resid else None
    ref = _ref_conv(xa, w, xb=xb, scale=scale, shift=shift, residual=res, ksize=k, stride=stride, pad=pad,
                    pad_mode=pad_mode, ups=ups, act=act)
    out = torch.full((B, Ho, Wo, Cout), float('nan'), device=dev)
    t = lambda v: None if v is None else v.to(dev)
    ops.conv2d(t(xa), t(w), out, src_b=t(xb), scale=t(scale), shift=t(shift), residual=t(res), ksize=k,
               stride=stride, pad=pad, pad_mode=pad_mode, upsample_a=ups, act=act, config=config)
    assert rel_err(out.cpu(), ref) < 2e-5, rel_err(out.cpu(), ref)



# B, H, W, Ca, Cb, Cout, stride, pad_mode, ups, act, resid, config
SPLITK_CASES = [
    (1, 6, 20, 128, 0, 32, 1, 0, False, 1, True, 22),     # layer4-like run tiles, 8 chunks -> 2 splits
    (2, 6, 20, 256, 0, 48, 1, 0, False, 1, False, 22),    # 16 chunks -> 4 splits, ragged Cout
    (1, 12, 40, 128, 128, 32, 1, 1, True, 2, False, 21),  # upconv_4_1-like: upsample + concat, reflect
    (1, 13, 21, 128, 0, 32, 2, 0, False, 1, False, 23),   # stride-2 stage entry
    (1, 8, 16, 144, 0, 16, 1, 0, False, 0, False, 20),    # 9 chunks over 2 splits (5 + 4)
]


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('case', SPLITK_CASES)
def test_conv2d_split_k(backend, case, monkeypatch):
    """Split-K of the LDS-patch kernel (clslam_conv_desc.workspace; the heuristic only splits 512-channel
    reductions, CLSLAM_SPLITK forces a split count): same result as torch, the scratch's
    arrival counters are back at zero afterwards (a second launch on the same scratch agrees bitwise),
    and the result matches the unsplit launch to fp32 reassociation."""
    dev = use_backend(backend)
    B, H, W, Ca, Cb, Cout, stride, pad_mode, ups, act, resid, config = case
    g = torch.Generator().manual_seed(11)
    Ha, Wa = (H // 2, W // 2) if ups else (H, W)
    xa = torch.randn(B, Ha, Wa, Ca, generator=g)
    xb = torch.randn(B, H, W, Cb, generator=g) if Cb else None
    w = torch.randn(Cout, 9, Ca + Cb, generator=g) * 0.05
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = torch.randn(B, Ho, Wo, Cout, generator=g) if resid else None
    ref = _ref_conv(xa, w, xb=xb, scale=scale, shift=shift, residual=res, ksize=3, stride=stride, pad=1, pad_mode=pad_mode,
                    ups=ups, act=act)
    t = lambda v: None if v is None else v.to(dev)   # noqa: E731
    ws = torch.zeros(4 << 20, dtype=torch.uint8, device=dev)
    monkeypatch.setattr(ops, '_CONV_WORKSPACES', {})     # engines of earlier tests register per-stream scratch: not here
    monkeypatch.setenv('CLSLAM_SPLITK', '3' if Ca + Cb >= 256 else '2')
    outs = []
    for workspace in (ws, ws, None):
        out = torch.full((B, Ho, Wo, Cout), float('nan'), device=dev)
        ops.conv2d(t(xa), t(w), out, src_b=t(xb), scale=t(scale), shift=t(shift), residual=t(res), ksize=3, stride=stride,
                   pad_mode=pad_mode, upsample_a=ups, act=act, config=config, workspace=workspace)
        outs.append(out.cpu())
    assert rel_err(outs[0], ref) < 2e-5
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0], outs[2]) < 1e-5 and not torch.equal(outs[0], outs[2])   # split really happened
    assert int(ws[:65536].view(torch.int32).abs().sum()) == 0


# B, H, W, Ca, Cb, Cout, stride, pad_mode, ups, act, resid, config, groups
STREAMK_CASES = [
    (2, 12, 20, 32, 0, 64, 1, 0, False, 1, True, 30, 5),       # 8x16 rect tiles, ragged image, 2 chunks/tile cut mid-tile
    (1, 10, 36, 64, 64, 64, 1, 1, True, 2, False, 30, 7),      # upsample + concat + reflect, 8 chunks
    (3, 6, 20, 48, 0, 128, 1, 0, False, 1, True, 31, 4),       # 4x16 rect, two channel tiles
    (1, 24, 40, 64, 0, 48, 1, 0, False, 0, False, 30, 256),    # ragged Cout, more groups than useful
    (2, 6, 20, 128, 0, 64, 1, 0, False, 1, True, 32, 6),       # run tiles: whole 6x20 image in one 128-px run
    (2, 12, 40, 64, 0, 128, 1, 0, False, 1, False, 32, 9),     # run tiles across rows (12x40)
    (1, 6, 20, 256, 0, 64, 1, 0, False, 1, False, 33, 16),     # ONE tile pair shared by 16 groups (B=1 regime)
    (2, 12, 40, 32, 0, 64, 2, 0, False, 1, False, 30, 3),      # stride 2
    (1, 9, 21, 32, 0, 64, 2, 0, False, 1, True, 31, 5),        # stride 2, odd sizes
    (2, 12, 40, 32, 0, 64, 2, 0, False, 1, True, 32, 5),       # stride-2 RUN tiles: a 6x20 output image is one 128-px run
    (3, 11, 37, 32, 0, 128, 2, 0, False, 1, False, 32, 4),     # ... odd input sizes (6x19 outputs), two channel tiles
    (1, 8, 12, 16, 0, 64, 2, 0, False, 0, False, 32, 2),       # ... a 4x6 output: several rows per run, mostly padding
    (1, 14, 44, 32, 0, 64, 1, 0, False, 0, False, 30, 11),     # dgrad-like padded domain (pad = 2 below)
    (2, 2, 4, 32, 0, 16, 1, 1, False, 2, False, 33, 3),        # tiny image, reflect
    # 256-thread groups x 32 channels (several groups per CU)
    (2, 12, 20, 32, 0, 64, 1, 0, False, 1, True, 34, 5),
    (1, 10, 36, 64, 64, 96, 1, 1, True, 2, False, 34, 7),
    (2, 12, 40, 32, 0, 64, 2, 0, False, 1, False, 34, 3),
    (3, 6, 20, 48, 0, 128, 1, 0, False, 1, True, 35, 13),
    (1, 9, 21, 32, 0, 64, 2, 0, False, 1, True, 35, 5),
    (2, 6, 20, 128, 0, 64, 1, 0, False, 1, True, 36, 6),
    (2, 12, 40, 64, 0, 48, 1, 0, False, 1, False, 37, 9),
]


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('case', STREAMK_CASES)
def test_conv2d_stream_k(backend, case, monkeypatch):
    """conv_sk.hip: persistent evenly-split implicit GEMM.  Same result as torch for every cut of the unit
    stream (forced through CLSLAM_SK_GROUPS), bitwise repeatable on the same scratch, hand-off flags back at
    zero afterwards."""
    dev = use_backend(backend)
    B, H, W, Ca, Cb, Cout, stride, pad_mode, ups, act, resid, config, groups = case
    g = torch.Generator().manual_seed(13)
    Ha, Wa = (H // 2, W // 2) if ups else (H, W)
    xa = torch.randn(B, Ha, Wa, Ca, generator=g)
    xb = torch.randn(B, H, W, Cb, generator=g) if Cb else None
    w = torch.randn(Cout, 9, Ca + Cb, generator=g) * 0.05
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    pad = 2 if (H, W) == (14, 44) else 1
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    res = torch.randn(B, Ho, Wo, Cout, generator=g) if resid else None
    ref = _ref_conv(xa, w, xb=xb, scale=scale, shift=shift, residual=res, ksize=3, stride=stride, pad=pad, pad_mode=pad_mode,
                    ups=ups, act=act)
    t = lambda v: None if v is None else v.to(dev)   # noqa: E731
    ws = torch.zeros(16 << 20, dtype=torch.uint8, device=dev)
    monkeypatch.setattr(ops, '_CONV_WORKSPACES', {})
    outs = []
    for grp in (groups, groups, 1):
        monkeypatch.setenv('CLSLAM_SK_GROUPS', str(grp))
        out = torch.full((B, Ho, Wo, Cout), float('nan'), device=dev)
        ops.conv2d(t(xa), t(w), out, src_b=t(xb), scale=t(scale), shift=t(shift), residual=t(res), ksize=3, stride=stride,
                   pad=pad, pad_mode=pad_mode, upsample_a=ups, act=act, config=config, workspace=ws)
        outs.append(out.cpu())
    assert rel_err(outs[0], ref) < 2e-5, rel_err(outs[0], ref)
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0], outs[2]) < 1e-5
    assert int(ws[:65536].view(torch.int32).abs().sum()) == 0
    with pytest.raises(Exception, match='workspace'):
        ops.conv2d(t(xa), t(w), out, src_b=t(xb), ksize=3, stride=stride, pad=pad, pad_mode=pad_mode, upsample_a=ups,
                   config=config, workspace=None)


@pytest.mark.parametrize('backend', BACKENDS)
def test_conv_profile_hook_times_every_launch(backend):
    """clslam_conv_profile_begin/end (bench.py's roofline leg): one kernel-exact duration per conv2d launch, in order."""
    dev = use_backend(backend)
    x = torch.randn(2, 12, 20, 32, device=dev)
    w = torch.randn(64, 9, 32, device=dev) * 0.1
    w1 = torch.randn(64, 1, 32, device=dev) * 0.1
    out = torch.empty(2, 12, 20, 64, device=dev)
    ops.profile_begin()
    ops.conv2d(x, w, out, ksize=3)
    ops.conv2d(x, w1, out, ksize=1, pad=0)
    ops.conv2d(x, w, out, ksize=3, config=21)
    got = ops.profile_end()
    assert ops.PROFILE is None and len(got) == 3
    assert [g[2] for g in got] == [2.0 * 480 * 64 * 9 * 32, 2.0 * 480 * 64 * 32, 2.0 * 480 * 64 * 9 * 32]
    assert got[2][1] == 21
    assert got[0][5] == 4.0 * (x.numel() + w.numel() + out.numel())      # algorithmic bytes: every operand once
    if backend == 'hip':
        assert all(1e-7 < g[3] < 1e-2 for g in got), got
    ops.conv2d(x, w, out, ksize=3)          # disarmed again: plain launches
    ops.profile_begin(); assert ops.profile_end() == []


# B, H, W, Cin, Cout, pad, act, resid, groups
WINO_CASES = [
    (1, 16, 16, 16, 64, 1, 1, True, 1),      # one 8x8-tile region, two stages, one workgroup
    (2, 12, 20, 32, 64, 1, 1, True, 5),      # 6x10 regions, ranges cut inside a region (hand-off), residual + ReLU
    (5, 6, 20, 32, 128, 1, 0, False, 7),     # two whole 6x20 images per region (last one half empty), two channel tiles
    (1, 24, 40, 16, 48, 1, 1, False, 3),     # ragged Cout (48 of a 64 tile), several regions per image
    (2, 7, 11, 32, 64, 1, 1, True, 4),       # odd image: half-valid tiles at the right / bottom edge
    (1, 14, 44, 32, 64, 2, 0, False, 6),     # dgrad-like padded domain (pad = 2: output 16x46)
    (3, 6, 20, 64, 64, 1, 1, True, 16),      # one region shared by many groups (B = 1 regime)
    (1, 48, 32, 16, 64, 1, 1, False, 2),     # tall image, 8x8 regions
]


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('config', [40])
@pytest.mark.parametrize('case', WINO_CASES)
def test_conv2d_winograd(backend, case, config, monkeypatch):
    """conv_wino.hip (config 40): F(2x2,3x3) on the 16x16x4 MFMA with the input transform in registers.  Same result as torch
    for every cut of the unit stream, bitwise repeatable on the same scratch (the per-wave hand-off flags hold the launch's epoch and
    are never reset: a second launch must not take the first one's flags for its own); fp32 Winograd carries about twice the
    rounding error of the direct form (3.6e-7...6.9e-7 of max|y| at 64...512 channels), far inside the 2e-5 of this file."""
    dev = use_backend(backend)
    B, H, W, Cin, Cout, pad, act, resid, groups = case
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, 9, Cin, generator=g) * 0.05
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    res = torch.randn(B, Ho, Wo, Cout, generator=g) if resid else None
    ref = _ref_conv(x, w, scale=scale, shift=shift, residual=res, ksize=3, stride=1, pad=pad, act=act)
    t = lambda v: None if v is None else v.to(dev)   # noqa: E731
    ws = torch.zeros(20 << 20, dtype=torch.uint8, device=dev)
    monkeypatch.setattr(ops, '_CONV_WORKSPACES', {})
    u = ops.wino_weight_transform(t(w))
    outs = []
    for grp in (groups, groups, 1):
        monkeypatch.setenv('CLSLAM_SK_GROUPS', str(grp))
        out = torch.full((B, Ho, Wo, Cout), float('nan'), device=dev)
        ops.conv2d(t(x), t(w), out, scale=t(scale), shift=t(shift), residual=t(res), ksize=3, pad=pad, act=act, config=config,
                   workspace=ws, weight_wino=u)
        outs.append(out.cpu())
    assert rel_err(outs[0], ref) < 2e-5, rel_err(outs[0], ref)
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0], outs[2]) < 1e-5
    with pytest.raises(Exception, match='weight_wino'):
        ops.conv2d(t(x), t(w), out, ksize=3, pad=pad, config=config, workspace=ws)



@pytest.mark.parametrize('backend', BACKENDS)
def test_auto_picked_winograd_falls_back_when_the_scratch_is_too_small(backend, monkeypatch):
    """clslam_conv2d_pick_config returns the Winograd kernel whenever a scratch pointer and a transformed filter are
    present; the kernel needs 64 KiB + 64 KiB per workgroup.  An AUTOMATICALLY picked Winograd launch that does not fit is served
    by the direct kernels (like an automatically picked stream-K configuration); an explicitly requested one still fails."""
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(2, 12, 20, 64, generator=g)
    w = torch.randn(64, 9, 64, generator=g) * 0.05
    ref = _ref_conv(x, w, ksize=3, stride=1, pad=1, act=1)
    t = lambda v: v.to(dev)   # noqa: E731
    monkeypatch.setattr(ops, '_CONV_WORKSPACES', {})
    u = ops.wino_weight_transform(t(w))
    small = torch.zeros(96 << 10, dtype=torch.uint8, device=dev)          # flags + HALF a slab
    d = ops.conv_desc(t(x), (2, 12, 20, 64), ksize=3)
    out = torch.full((2, 12, 20, 64), float('nan'), device=dev)
    ops.conv2d(t(x), t(w), out, ksize=3, act=1, workspace=small, weight_wino=u, cu_limit=2)
    assert rel_err(out.cpu(), ref) < 2e-5
    with pytest.raises(Exception, match='workspace'):
        ops.conv2d(t(x), t(w), out, ksize=3, act=1, workspace=small, weight_wino=u, cu_limit=2, config=40)
    del d


@pytest.mark.parametrize('backend', BACKENDS)
def test_conv2d_call_site_descriptor_cache(backend):
    """conv2d(..., key=...): a fixed call site's descriptor is re-used only while source, weight and output sit where they sat --
    another tensor under the same key gets a descriptor of its own, the result is the uncached call's every time."""
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(16, 9, 16, generator=g) / 12).to(dev)
    b = torch.randn(16, generator=g).to(dev)
    key = ('test-site', backend)
    ops._CONV_DESC_CACHE.pop(key, None)
    xs = [torch.randn(1, 8, 16, 16, generator=g).to(dev) for _ in range(2)]
    for rnd in range(3):
        for x in xs:                      # the SAME key sees two different sources (and fresh outputs): validated, not trusted
            ref = torch.empty(1, 8, 16, 16, device=dev)
            ops.conv2d(x, w, ref, shift=b, ksize=3, act=2)
            out = torch.full((1, 8, 16, 16), float('nan'), device=dev)
            ops.conv2d(x, w, out, shift=b, ksize=3, act=2, key=key)
            assert torch.equal(out, ref)
            out2 = out.clone().fill_(float('nan'))
            d_before = ops._CONV_DESC_CACHE[key][0]
            ops.conv2d(x, w, out2, shift=b, ksize=3, act=2, key=key)      # other output buffer: rebuilt
            assert torch.equal(out2, ref) and ops._CONV_DESC_CACHE[key][0] is not d_before
            d_before = ops._CONV_DESC_CACHE[key][0]
            out2.fill_(float('nan'))
            ops.conv2d(x, w, out2, shift=b, ksize=3, act=2, key=key)      # same three tensors: the kept descriptor
            assert torch.equal(out2, ref) and ops._CONV_DESC_CACHE[key][0] is d_before
