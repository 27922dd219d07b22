#!/usr/bin/env python
"""SURVEY.md 8(d)'s end-to-end frame, a few of them back to back, for a rocprofv3 timeline (tools/timeline_e2e.py) and for host-side
stamps: pinned host minibatch -> adapt() (H2D inside) -> cam_T_cam[0] + every loss scalar on the host (slam.py:181-188).

    python tools/e2e_frames.py [frames=12] [replay=4]        prints ms per frame (median of the last frames) + host phase stamps"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'cl-slam_amd'))
import torch  # noqa: E402
import bench  # noqa: E402
from clslam_hip import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W, B = 192, 640, 1 + R
dev = torch.device('cuda:0')
torch.manual_seed(1)
p = bench.build_predictor(H, W, B)
full = synth.make_batch(B, H, W, seed=0)
host = {k: v.pin_memory() for k, v in full.items()}
eng = p.engine
state = None


def frame(stamps=None):
    t0 = time.perf_counter()
    out, losses = p.adapt(None, dict(host), steps=1)
    t1 = time.perf_counter()
    T = out['cam_T_cam', 0, 1][0, :].squeeze().cpu().detach().numpy()
    t2 = time.perf_counter()
    vals = {k: float(v.squeeze().cpu().detach().numpy()) for k, v in losses.items()}
    t3 = time.perf_counter()
    if stamps is not None:
        stamps.append((t0, t1, t2, t3))
    return T, vals


for i in range(6):
    frame()
torch.cuda.synchronize()
state = (eng.w.clone(), eng.m.clone(), eng.v.clone(), eng.adam_step_count)
stamps = []
torch.cuda.synchronize()
t_begin = time.perf_counter()
for i in range(N):
    frame(stamps)
torch.cuda.synchronize()
t_end = time.perf_counter()
print(f'# {N} end-to-end frames, K={R}: {(t_end - t_begin) / N * 1e3:.3f} ms per frame')
print('# host stamps per frame (ms from the frame\'s adapt() call): adapt() returned | pose on host | losses on host | next frame starts')
for i, (t0, t1, t2, t3) in enumerate(stamps):
    nxt = stamps[i + 1][0] if i + 1 < len(stamps) else t_end
    print(f'  frame {i:2d}: {1e3 * (t1 - t0):6.3f} | {1e3 * (t2 - t0):6.3f} | {1e3 * (t3 - t0):6.3f} | {1e3 * (nxt - t0):6.3f}')
