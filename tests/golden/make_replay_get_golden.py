#!/usr/bin/env python
"""Generates tests/golden/replay_get.npz by running the REFERENCE's own ``ReplayBuffer._get`` (slam/replay_buffer.py:263-291)
in the build container: synthetic PNG frames + pickled samples on disk, `random.seed(seed)`, then `_get(filename)` per sample.
Third-party packages the container lacks are stubbed (tests/ref_stubs.py); torchvision's tensor-path adjust_* functions are the
restatement in oracle/jitter_tensor.py (torchvision itself is not installable here: that part is PARITY UNPINNED), Resize /
ToTensor run on the real Pillow.  The fixture holds DATA only: raw uint8 frames, the seed, and every tensor of the returned dicts.

    python tests/golden/make_replay_get_golden.py
"""
import pickle
import random
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

TESTS = Path(__file__).resolve().parents[1]
ROOT = TESTS.parent
REF = Path('/root/reference')
H, W, RAW_H, RAW_W, SEED, N = 32, 64, 40, 84, 1234, 2


def main() -> None:
    sys.path[:0] = [str(ROOT / 'cl-slam_amd'), str(REF)]
    sys.path += [str(TESTS), str(ROOT)]
    import ref_stubs
    ref_stubs.install()
    from emu_util import use_backend
    use_backend('emu')                      # ReplayBuffer.__init__ builds the loop-closure FeatureEncoder
    import os
    from test_lcd_encoder import _weights
    from PIL import Image
    work = Path(tempfile.mkdtemp())
    torch.save(_weights()[1], work / 'mbv3.pth')
    os.environ['CLSLAM_MOBILENETV3_WEIGHTS'] = str(work / 'mbv3.pth')
    import slam.replay_buffer as rb
    assert str(Path(rb.__file__).resolve()).startswith(str(REF))
    buf = rb.ReplayBuffer(work / 'buffer', 'Kitti', None, H, W, [0, 1, 2, 3], [0, -1, 1], batch_size=2, maximize_diversity=True,
                          max_buffer_size=2, similarity_threshold=0.9999, do_augmentation=True)
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:RAW_H, 0:RAW_W].astype(np.float32)
    out = {'height': H, 'width': W, 'scales': np.array([0, 1, 2, 3]), 'frames': np.array([0, -1, 1]), 'seed': SEED, 'n_samples': N,
           'camera_matrix': np.eye(4, dtype=np.float32)[None]}
    files = []
    for i in range(N):
        sample = {('camera_matrix', 0): torch.from_numpy(out['camera_matrix']).clone(), ('index',): torch.tensor([i])}
        for f in (0, -1, 1):
            img = np.zeros((RAW_H, RAW_W, 3), np.float32)
            for _ in range(5):
                fx, fy, ph = rng.uniform(0.05, 0.5), rng.uniform(0.05, 0.6), rng.uniform(0, 6.28, 3)
                for c in range(3):
                    img[..., c] += np.sin(fx * xx + fy * yy + ph[c])
            img = (img - img.min()) / (img.max() - img.min())
            img = (img * 255).astype(np.uint8)
            img[:4, :6] = rng.integers(0, 256, (4, 6, 1))         # a gray patch and hard edges
            png = work / f's{i}_f{f}.png'
            Image.fromarray(img).save(png)
            out[f'raw_{i}_{f}'] = img
            sample['rgb', f] = png
        fn = work / f'kitti_{i:05}.pkl'
        with open(fn, 'wb') as fh:
            pickle.dump(sample, fh)
        files.append(fn)
    random.seed(SEED)
    for i, fn in enumerate(files):
        data = buf._get(fn)                                      # THE reference code path
        for k, v in data.items():
            out['|'.join([f'out{i}'] + [str(p) for p in k])] = v.numpy()
    np.savez_compressed(TESTS / 'golden' / 'replay_get.npz', **out)
    print('wrote', TESTS / 'golden' / 'replay_get.npz', sum(v.nbytes for v in out.values() if hasattr(v, 'nbytes')) // 1024, 'KiB raw')


if __name__ == '__main__':
    main()
