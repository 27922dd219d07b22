#!/usr/bin/env python
"""Soak run: N end-to-end frames (pinned host minibatch -> adapt() -> pose + losses on the host, a descriptor pass every frame
like slam.py:143-147) with fresh host dicts and changing content; reports time per frame and the allocator's footprint per
block of frames -- device memory must be flat (the upload path allocates per frame and defers reuse with record_stream).
    python tools/soak.py [frames=2000] [replay=4]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'cl-slam_amd')); sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from clslam_hip import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W, B = 192, 640, 1 + K
p = bench.build_predictor(H, W, B)
p.optimizer.param_groups[0]['lr'] = 1e-6            # a long run on one synthetic scene: keep the untrained network from collapsing
hosts = [{k: v.pin_memory() for k, v in synth.make_batch(B, H, W, seed=s).items()} for s in range(4)]
dev = p.device
t0 = time.perf_counter()
worst = 0.0
for f in range(N):
    host = dict(hosts[f % 4])
    x = host['rgb', 0, 0][:1].to(dev)
    feat = p.models['depth_encoder'](x)[4].mean(-1).mean(-1).cpu()
    out, losses = p.adapt(None, host, steps=1)
    T = out['cam_T_cam', 0, 1][0, :].squeeze().cpu().numpy()
    loss = float(losses['loss'])
    assert loss == loss and abs(T).max() < 1e3
    if (f + 1) % (N // 4) == 0:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (N // 4) * 1e3
        print(f'frames {f + 1 - N // 4:5d}..{f + 1:5d}: {dt:.3f} ms/frame  loss {loss:.5f}  allocated {torch.cuda.memory_allocated() / 2**20:8.1f} MiB  '
              f'reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB  peak {torch.cuda.max_memory_allocated() / 2**20:8.1f} MiB', flush=True)
        t0 = time.perf_counter()
print('soak ok')
