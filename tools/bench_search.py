#!/usr/bin/env python
"""Latency of FlatIPIndex.search at the reference's sizes, next to numpy on the host."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'cl-slam_amd'))
from clslam_hip.flat_index import FlatIPIndex, normalize_L2  # noqa: E402

for n, d, k, what in ((4000, 576, 100, 'loop closure: 4000 frames x 576, top 100'), (100, 512, 1, 'replay buffer: nearest of 100 x 512'),
                      (100, 512, 100, 'replay buffer: full ranking of 100 x 512')):
    x = np.random.default_rng(0).standard_normal((n, d)).astype(np.float32)
    normalize_L2(x)
    idx = FlatIPIndex(d)
    idx.add(x)
    q = x[:1].copy()
    for _ in range(5):
        idx.search(q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        idx.search(q, k)
    t_gpu = (time.perf_counter() - t0) / 50 * 1e6
    t0 = time.perf_counter()
    for _ in range(50):
        s = x @ q[0]
        np.argsort(-s, kind='stable')[:k]
    t_np = (time.perf_counter() - t0) / 50 * 1e6
    print(f'{what:48s} GPU {t_gpu:7.1f} us per search (incl. D2H of the result)   numpy on the host {t_np:7.1f} us')
