#!/usr/bin/env python
"""Latency of the inference-only entry points (BASELINE config 1 shape: one 192x640 triplet): predict(),
adapt(online, None), predict_pose(), models['depth_encoder'](img) -- synchronised per call, inputs on the GPU."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'cl-slam_amd'))
import bench  # noqa: E402
from clslam_hip import synth  # noqa: E402

H, W = 192, 640
p = bench.build_predictor(H, W, 1)
batch = {k: v.cuda() for k, v in synth.make_batch(1, H, W, seed=0).items()}
img0, img1 = batch['rgb', 0, 0][0], batch['rgb', 1, 0][0]


def timeit(name, fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()
    print(f'{name:44s} {(time.perf_counter() - t0) / n * 1e3:7.3f} ms per call (synchronised)')


timeit('predict(batch), B=1', lambda: p.predict(batch))
timeit('adapt(online, None), B=1', lambda: p.adapt(batch, None))
timeit('predict_pose(img0, img1, as_numpy=True)', lambda: p.predict_pose(img0, img1))
timeit("models['depth_encoder'](img)[4] (slam.py:146)", lambda: p.models['depth_encoder'](batch['rgb', 0, 0])[4])
