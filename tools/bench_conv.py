#!/usr/bin/env python
"""Micro-benchmark of clslam_conv2d / clslam_conv_wgrad on the real layer shapes (192x640, B=5).
Prints achieved TFLOP/s per layer and tile configuration (fp32 MFMA peak = 157.3)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'cl-slam_amd'))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import _variant  # noqa: F401,E402  (CLSLAM_TOOL_LIB=<tag>: a probe build of the library, tools/build_variant.py)
from clslam_hip import ops  # noqa: E402

dev = torch.device('cuda:0')
SPLITK = bool(int(__import__('os').environ.get('CLSLAM_SPLITK', '0') or 0))
WGRAD = __import__('os').environ.get('BENCH_WGRAD', '1') != '0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H, W = 192, 640
# name, B mult, Hi, Wi, Ca, Cb, Cout, k, stride, reflect, ups
LAYERS = [
    ('layer1 64->64 @48x160', 1, 48, 160, 64, 0, 64, 3, 1, 0, 0),
    ('layer2 128->128 @24x80', 1, 24, 80, 128, 0, 128, 3, 1, 0, 0),
    ('layer3 256->256 @12x40', 1, 12, 40, 256, 0, 256, 3, 1, 0, 0),
    ('layer4 512->512 @6x20', 1, 6, 20, 512, 0, 512, 3, 1, 0, 0),
    ('layer4 pose 2B', 2, 6, 20, 512, 0, 512, 3, 1, 0, 0),
    ('layer3 pose 2B', 2, 12, 40, 256, 0, 256, 3, 1, 0, 0),
    ('layer3.0 128->256 s2', 1, 24, 80, 128, 0, 256, 3, 2, 0, 0),
    ('layer4.0 256->512 s2', 1, 12, 40, 256, 0, 512, 3, 2, 0, 0),
    ('pose dec 256->256 2B', 2, 6, 20, 256, 0, 256, 3, 1, 0, 0),
    ('upconv_4_0 512->256 @6x20', 1, 6, 20, 512, 0, 256, 3, 1, 1, 0),
    ('upconv_3_0 256->128 @12x40', 1, 12, 40, 256, 0, 128, 3, 1, 1, 0),
    ('layer2.0 64->128 s2', 1, 48, 160, 64, 0, 128, 3, 2, 0, 0),
    ('upconv_4_1 512->256 @12x40', 1, 12, 40, 256, 256, 256, 3, 1, 1, 1),
    ('upconv_3_1 256->128 @24x80', 1, 24, 80, 128, 128, 128, 3, 1, 1, 1),
    ('upconv_2_1 128->64 @48x160', 1, 48, 160, 64, 64, 64, 3, 1, 1, 1),
    ('upconv_1_1 96->32 @96x320', 1, 96, 320, 32, 64, 32, 3, 1, 1, 1),
    ('upconv_0_0 32->16 @96x320', 1, 96, 320, 32, 0, 16, 3, 1, 1, 0),
    ('upconv_0_1 16->16 @192x640', 1, 192, 640, 16, 0, 16, 3, 1, 1, 1),
    ('upconv_2_0 128->64 @24x80', 1, 24, 80, 128, 0, 64, 3, 1, 1, 0),
    ('upconv_1_0 64->32 @48x160', 1, 48, 160, 64, 0, 32, 3, 1, 1, 0),
]


# data-gradient convolutions of the depth decoder: flipped weights on the padded domain (pad = 2: the output is (H+2) x (W+2))
DGRAD_LAYERS = [
    ('dgrad 256->256 @12x40 pad2', 1, 12, 40, 256, 0, 256, 3, 1, 0, 0, 2),
    ('dgrad 128->256 @12x40 pad2', 1, 12, 40, 128, 0, 256, 3, 1, 0, 0, 2),
    ('dgrad 128->128 @24x80 pad2', 1, 24, 80, 128, 0, 128, 3, 1, 0, 0, 2),
    ('dgrad 64->128 @24x80 pad2', 1, 24, 80, 64, 0, 128, 3, 1, 0, 0, 2),
    ('dgrad 64->64 @48x160 pad2', 1, 48, 160, 64, 0, 64, 3, 1, 0, 0, 2),
    ('dgrad 32->64 @48x160 pad2', 1, 48, 160, 32, 0, 64, 3, 1, 0, 0, 2),
    ('dgrad 256->256 @6x20 pose 2B', 2, 6, 20, 256, 0, 256, 3, 1, 0, 0, 1),
]
if __import__('os').environ.get('BENCH_DGRAD'):
    LAYERS = DGRAD_LAYERS


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


_sel = __import__('os').environ.get('BENCH_LAYERS')
if _sel:
    LAYERS = [LAYERS[int(i)] for i in _sel.split(',')]
for name, bm, Hi, Wi, Ca, Cb, Cout, k, stride, refl, ups, *rest in LAYERS:
    pad = rest[0] if rest else k // 2
    Bn = B * bm
    Ha, Wa = (Hi // 2, Wi // 2) if ups else (Hi, Wi)
    xa = torch.randn(Bn, Ha, Wa, Ca, device=dev)
    xb = torch.randn(Bn, Hi, Wi, Cb, device=dev) if Cb else None
    w = torch.randn(Cout, k * k, Ca + Cb, device=dev) * 0.05
    Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
    out = torch.empty(Bn, Ho, Wo, Cout, device=dev)
    flops = 2.0 * Bn * Ho * Wo * Cout * k * k * (Ca + Cb)
    line = f'{name:32s} M={Bn*Ho*Wo:7d} {flops/1e9:7.2f} GF |'
    cfgs = [-1] + ([int(c) for c in sys.argv[2].split(',')] if len(sys.argv) > 2 else [])
    wsk = torch.zeros(32 << 20, dtype=torch.uint8, device=dev)
    ref_out = None
    wino_ok = k == 3 and stride == 1 and not Cb and not ups and not refl
    u = ops.wino_weight_transform(w) if (40 in cfgs and wino_ok) else None
    for cfg in cfgs:
        if cfg == 40 and not wino_ok:
            line += f' c{cfg}:   -  '
            continue
        try:
            ws_arg = wsk if cfg >= 30 else None       # stream-K configs need the zero-filled scratch
            t = timeit(lambda: ops.conv2d(xa, w, out, src_b=xb, ksize=k, stride=stride, pad=pad, pad_mode=refl, upsample_a=bool(ups),
                                          act=1, config=cfg, workspace=ws_arg, weight_wino=u if cfg == 40 else None))
            line += f' c{cfg}:{flops/t/1e12:6.1f}'
            if cfg == -1:
                ref_out = out.clone()
            elif ref_out is not None:
                err = float((out - ref_out).abs().max() / ref_out.abs().max())
                if not err < 1e-5:
                    line += f'(ERR {err:.1e})'
            if cfg == -1 and SPLITK:
                t = timeit(lambda: ops.conv2d(xa, w, out, src_b=xb, ksize=k, stride=stride, pad_mode=refl, upsample_a=bool(ups),
                                              act=1, config=cfg, workspace=wsk))
                line += f' splitK:{flops/t/1e12:6.1f}'
        except Exception:
            line += f' c{cfg}:   -  '
    if not WGRAD or rest:
        print(line, flush=True)
        continue
    # wgrad
    dz = torch.randn(Bn, Ho, Wo, Cout, device=dev)
    desc = ops.conv_desc(xa, (Bn, Ho, Wo, Cout), src_b=xb, ksize=k, stride=stride, pad_mode=refl, upsample_a=bool(ups))
    for target in (512, 2048):
        splits = ops.wgrad_splits(desc, target)
        part = torch.empty(splits * w.numel(), device=dev)
        dw = torch.empty(w.numel(), device=dev)

        def f():
            ops.conv_wgrad(desc, dz, part, splits)
            ops.reduce_partials(part, dw, w.numel(), splits)
        t = timeit(f)
        tk = timeit(lambda: ops.conv_wgrad(desc, dz, part, splits))
        tr = timeit(lambda: ops.reduce_partials(part, dw, w.numel(), splits))
        line += f' | wg{target}(s{splits}):{flops/t/1e12:6.1f} [{tk*1e6:.0f}+{tr*1e6:.0f}us]'
    if ops.wgrad_patch_supported(desc):
        for target in (512, 1024, 2048):
            splits = ops.wgrad_patch_splits(desc, target)
            part = torch.empty(splits * w.numel(), device=dev)

            def f2():
                ops.conv_wgrad_patch(desc, dz, part, splits)
                ops.reduce_partials(part, dw, w.numel(), splits)
            t = timeit(f2)
            tk = timeit(lambda: ops.conv_wgrad_patch(desc, dz, part, splits))
            tr = timeit(lambda: ops.reduce_partials(part, dw, w.numel(), splits))
            line += f' | wgP{target}(s{splits}):{flops/t/1e12:6.1f} [{tk*1e6:.0f}+{tr*1e6:.0f}us]'
    print(line, flush=True)
