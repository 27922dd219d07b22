#!/usr/bin/env python
"""Pin oracle/jitter_tensor.py -- the restatement of torchvision's FLOAT-TENSOR colour jitter the replay buffer applies to every
replayed sample (slam/replay_buffer.py:264-283 -> datasets/utils.py:236-259 -> torchvision.transforms.functional.adjust_*) --
against the REAL torchvision, the moment a build container has one (requirements.txt:6 pins 0.11.1).

    python tests/golden/make_replay_jitter_golden.py

With torchvision importable:
  1. every op alone, all 24 orders of the four ops and the edge images of tests/test_replay_ingest.py (gray rows, saturated red, exact
     byte fractions, black) go through torchvision.transforms.functional.adjust_brightness / _contrast / _saturation / _hue
     (tensor inputs) and through the restatement: they must agree BITWISE where no op follows `contrast`, to 1e-6 otherwise;
  2. tests/golden/replay_jitter.npz is written: the input parameters (seeds, orders, factors) and torchvision's outputs -- data,
     no source -- for tests/test_replay_ingest.py::test_jitter_restatement_matches_torchvision_golden.
Without torchvision it prints PARITY UNPINNED and exits 3: nothing is faked."""
import itertools
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
OUT = Path(__file__).resolve().parent
for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
    sys.path.insert(0, str(p))


def images() -> torch.Tensor:
    """the image set of tests/test_replay_ingest.py::test_float_tensor_jitter_matches_the_torchvision_restatement"""
    g = torch.Generator().manual_seed(3)
    imgs = torch.rand(6, 3, 20, 36, generator=g)
    imgs[1, :, :5] = imgs[1, :1, :5]
    imgs[2, 0] = 1.0
    imgs[3] = (imgs[3] * 255).round() / 255
    imgs[4] = 0.0
    return imgs


def cases():
    c = [([op], [1.15, 0.85, 1.2, -0.1]) for op in range(4)] + [([op], [0.8, 1.2, 0.8, 0.1]) for op in range(4)]
    c += [(list(perm), [0.93, 1.07, 1.13, 0.037]) for perm in itertools.permutations(range(4))]
    c += [([], [1, 1, 1, 0]), ([3], [1, 1, 1, -0.5]), ([3], [1, 1, 1, 0.5]), ([1, 3], [1, 0.8, 1, 0.0])]
    return c


def main() -> int:
    try:
        import torchvision
        from torchvision.transforms import functional as F
    except Exception as e:  # noqa: BLE001
        print(f'PARITY UNPINNED: torchvision is not importable here ({e}); oracle/jitter_tensor.py stays a restatement of the published '
              '0.11.1 functional_tensor.py, checked against the HIP path and against the reference\'s own _get flow only.  Re-run this '
              'script in a container with torchvision to pin it.')
        return 3
    from oracle import jitter_tensor as jt
    print('torchvision', torchvision.__version__)
    ops = (F.adjust_brightness, F.adjust_contrast, F.adjust_saturation, F.adjust_hue)
    imgs = images()
    outs, worst, worst_bitwise = [], 0.0, 0.0
    for order, factors in cases():
        ref = []
        for i in range(len(imgs)):                       # one image at a time: the replay buffer jitters (1, 3, h, w) tensors
            x = imgs[i:i + 1]
            for op in order:
                x = ops[op](x, float(factors[op]))
            ref.append(x)
        ref = torch.cat(ref)
        got = torch.cat([jt.color_jitter(imgs[i:i + 1], order, factors) for i in range(len(imgs))])
        err = float((got - ref).abs().max())
        after_contrast = 1 in order and order.index(1) < len(order) - 1
        if after_contrast:
            worst = max(worst, err)
        else:
            worst_bitwise = max(worst_bitwise, err)
        outs.append(ref.numpy())
    print(f'restatement vs torchvision: max |diff| {worst_bitwise:.2e} where no op follows contrast (must be 0), {worst:.2e} otherwise')
    assert worst_bitwise == 0.0 and worst <= 1e-6, (worst_bitwise, worst)
    np.savez_compressed(OUT / 'replay_jitter.npz', torchvision_version=np.array(torchvision.__version__),
                        orders=np.array([o + [-1] * (4 - len(o)) for o, _ in cases()], dtype=np.int32),
                        factors=np.array([f for _, f in cases()], dtype=np.float64), outputs=np.stack(outs))
    print('PINNED: wrote', OUT / 'replay_jitter.npz')
    return 0


if __name__ == '__main__':
    sys.exit(main())
