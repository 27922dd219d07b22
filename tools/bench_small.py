#!/usr/bin/env python
"""Time the non-conv kernels of the adapt step one by one on the real 192x640 B=5 workspace."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'cl-slam_amd')); sys.path.insert(0, str(ROOT))
import torch
import bench
from clslam_hip import ops, synth

H, W, B = 192, 640, 5
p = bench.build_predictor(H, W, B)
eng = p.engine
batch = {k: v.to(p.device) for k, v in synth.make_batch(B, H, W, seed=0).items()}
for _ in range(2):
    p.adapt(None, batch, steps=1)
torch.cuda.synchronize()
ws = eng._ws[B]


def timeit(name, fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:34s} {e0.elapsed_time(e1) / iters * 1e3:8.1f} us', flush=True)


aug = {f: batch['rgb_aug', f, 0].contiguous() for f in (-1, 0, 1)}
e = eng.enc['depth_encoder']
pe = eng.enc['pose_encoder']
timeit('stem depth (3ch, B)', lambda: ops.stem_conv(aug[0], None, e.stem_w, e.stem_scale, e.stem_shift, ws.denc.f0))
timeit('stem pose (6ch, B)', lambda: ops.stem_conv(aug[-1], aug[0], pe.stem_w, pe.stem_scale, pe.stem_shift, ws.penc.f0[:B]))
timeit('maxpool depth', lambda: ops.maxpool3x3s2(ws.denc.f0, ws.denc.pool))
timeit('maxpool pose (2B)', lambda: ops.maxpool3x3s2(ws.penc.f0, ws.penc.pool))
