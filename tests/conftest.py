import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT / 'cl-slam_amd', ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the oracle is torch on the CPU: one thread per hardware thread is far from its best setting on a many-core host (the
    # MI355X box: 3.7 s per B = 5 step with 128 threads, 0.33 s with 16 -- bench.py's cpu_baseline sweep), and the parity tests
    # spend most of their time in it
    import torch
    torch.set_num_threads(min(16, torch.get_num_threads()))


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
