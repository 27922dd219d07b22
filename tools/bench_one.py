#!/usr/bin/env python
"""Run ONE conv layer/config a few times (for rocprofv3 --pmc counter collection).
    python tools/bench_one.py <layer-substring> <config> [B]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'cl-slam_amd'))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from clslam_hip import ops  # noqa: E402

LAYERS = {
    'layer1': (1, 48, 160, 64, 0, 64, 3, 1, 0, 0),
    'layer2': (1, 24, 80, 128, 0, 128, 3, 1, 0, 0),
    'layer3': (1, 12, 40, 256, 0, 256, 3, 1, 0, 0),
    'layer4': (2, 6, 20, 512, 0, 512, 3, 1, 0, 0),
    'upconv_2_1': (1, 48, 160, 64, 64, 64, 3, 1, 1, 1),
    'upconv_1_1': (1, 96, 320, 32, 64, 32, 3, 1, 1, 1),
    'upconv_0_1': (1, 192, 640, 16, 0, 16, 3, 1, 1, 1),
}
name, cfg = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 5
bm, Hi, Wi, Ca, Cb, Cout, k, stride, refl, ups = LAYERS[name]
dev = torch.device('cuda:0')
Bn = B * bm
Ha, Wa = (Hi // 2, Wi // 2) if ups else (Hi, Wi)
xa = torch.randn(Bn, Ha, Wa, Ca, device=dev)
xb = torch.randn(Bn, Hi, Wi, Cb, device=dev) if Cb else None
w = torch.randn(Cout, k * k, Ca + Cb, device=dev) * 0.05
out = torch.empty(Bn, Hi, Wi, Cout, device=dev)
flops = 2.0 * Bn * Hi * Wi * Cout * k * k * (Ca + Cb)
for _ in range(5):
    ops.conv2d(xa, w, out, src_b=xb, ksize=k, stride=stride, pad_mode=refl, upsample_a=bool(ups), act=1, config=cfg)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10):
    ops.conv2d(xa, w, out, src_b=xb, ksize=k, stride=stride, pad_mode=refl, upsample_a=bool(ups), act=1, config=cfg)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print(f'{name} cfg {cfg}: {flops / t / 1e12:.1f} TFLOP/s, {t * 1e6:.1f} us')
