#!/usr/bin/env python
"""Pin oracle/mobilenet.py against the REAL torchvision ``mobilenet_v3_small`` the reference builds
(loop_closure_detection/encoder.py:13-33) -- the moment a build container has torchvision (requirements.txt:6 pins
0.11.1) and, for the second half, the ImageNet weights the reference downloads.

    python tests/golden/make_lcd_golden.py [--weights mobilenet_v3_small-047dcff4.pth]

What it does when torchvision is importable:
  1. architecture pin: torchvision's network and the oracle's restatement get the SAME closed-form state dict
     (tests/test_lcd_encoder.py::_weights); their 'flatten' features on seeded images must agree to 1e-6.  A key or
     shape mismatch between the two state dicts is a failure by itself.
  2. if the ImageNet checkpoint is available (--weights / $CLSLAM_MOBILENETV3_WEIGHTS / torch hub cache): the same
     comparison with the real weights, and tests/golden/lcd_features.npz is written (seeded inputs' parameters +
     torchvision's 576-d outputs -- data, no source) for tests/test_lcd_encoder.py::test_oracle_matches_torchvision_golden.
Without torchvision it prints PARITY UNPINNED and exits 3: nothing is faked."""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
OUT = Path(__file__).resolve().parent
for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
    sys.path.insert(0, str(p))


def seeded_images(n: int, H: int, W: int, seed: int) -> torch.Tensor:
    from clslam_hip import synth
    return synth.make_batch(n, H, W, seed=seed)['rgb', 1, 0]


def torchvision_features(tv_model, img: torch.Tensor) -> torch.Tensor:
    """exactly loop_closure_detection/encoder.py:22-33"""
    from torchvision import transforms
    from torchvision.models.feature_extraction import create_feature_extractor
    fx = create_feature_extractor(tv_model, return_nodes=['flatten']).eval()
    norm = transforms.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
    with torch.no_grad():
        return fx(norm(img))['flatten']


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument('--weights', default=None)
    args = ap.parse_args()
    try:
        import torchvision
        from torchvision import models
    except Exception as e:  # noqa: BLE001
        print(f'PARITY UNPINNED: torchvision is not importable here ({e}); oracle/mobilenet.py stays checked against itself '
              'and the HIP path only.  Re-run this script in a container with torchvision to pin it.')
        return 3
    from oracle.mobilenet import MobileNetV3SmallFeatures, feature_encoder
    from test_lcd_encoder import _weights
    print('torchvision', torchvision.__version__)
    oracle, sd = _weights()
    tv = models.mobilenet_v3_small()
    tv_sd = tv.state_dict()
    feat_keys = {k: v.shape for k, v in tv_sd.items() if k.startswith('features.')}
    mine = {k: v.shape for k, v in sd.items()}
    assert feat_keys == mine, {'only_torchvision': sorted(set(feat_keys) - set(mine))[:5], 'only_oracle': sorted(set(mine) - set(feat_keys))[:5],
                               'shape': [k for k in feat_keys if k in mine and feat_keys[k] != mine[k]][:5]}
    tv.load_state_dict({**tv_sd, **sd})
    worst = 0.0
    for (n, H, W, seed) in ((2, 64, 128, 6), (1, 192, 640, 7), (1, 224, 224, 8)):
        img = seeded_images(n, H, W, seed)
        a, b = torchvision_features(tv, img), feature_encoder(oracle, img)
        err = float((a - b).abs().max() / a.abs().max())
        worst = max(worst, err)
        print(f'closed-form weights {n}x3x{H}x{W}: rel err {err:.2e}')
    assert worst < 1e-6, worst
    print('ARCHITECTURE PINNED against torchvision', torchvision.__version__)
    cands = [args.weights, os.environ.get('CLSLAM_MOBILENETV3_WEIGHTS'),
             Path(torch.hub.get_dir()) / 'checkpoints' / 'mobilenet_v3_small-047dcff4.pth']
    wfile = next((Path(c) for c in cands if c and Path(c).exists()), None)
    if wfile is None:
        print("ImageNet weights not found: PARITY WITH THE REFERENCE'S PRETRAINED ENCODER STAYS UNPINNED (architecture pinned).")
        return 2
    real = torch.load(wfile, map_location='cpu')
    tv.load_state_dict(real)
    oracle = MobileNetV3SmallFeatures()
    oracle.load_state_dict({k: v for k, v in real.items() if k.startswith('features.')})
    out = {}
    for i, (n, H, W, seed) in enumerate(((2, 64, 128, 6), (1, 192, 640, 7))):
        img = seeded_images(n, H, W, seed)
        a, b = torchvision_features(tv, img), feature_encoder(oracle.eval(), img)
        err = float((a - b).abs().max() / a.abs().max())
        print(f'ImageNet weights {n}x3x{H}x{W}: rel err {err:.2e}')
        assert err < 1e-6, err
        out[f'params_{i}'] = np.array([n, H, W, seed], np.int64)
        out[f'features_{i}'] = a.numpy()
    out['weights_sha1_head'] = np.frombuffer(__import__('hashlib').sha1(wfile.read_bytes()).digest()[:8], np.uint8)
    np.savez_compressed(OUT / 'lcd_features.npz', **out)
    print('PINNED; wrote', OUT / 'lcd_features.npz')
    return 0


if __name__ == '__main__':
    sys.exit(main())
