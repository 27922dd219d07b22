#!/usr/bin/env python
"""Generates tests/golden/resize_*.npz with the REAL Pillow of the build container (the reference's
datasets resize through torchvision.transforms.Resize(LANCZOS) = PIL.Image.resize, datasets/utils.py:62-66,
154-163).  Inputs are seeded uint8 noise + a smooth ramp; outputs are Pillow's.

    python tests/golden/make_resize_golden.py
"""
from pathlib import Path

import numpy as np
import PIL
from PIL import Image

OUT = Path(__file__).resolve().parent


def image(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    ramp = (((x * 5 + y * 3) % 256)[..., None] * np.array([1, 0.5, 0.25])).astype(np.int64)
    noise = rng.integers(0, 256, (h, w, 3))
    return ((ramp + noise) // 2 % 256).astype(np.uint8)


def pil_resize(img, oh, ow):
    return np.asarray(Image.fromarray(img).resize((ow, oh), Image.LANCZOS))


cases = {}
# single resizes: integer / fractional downscale, odd sizes, upscale, one axis unchanged
for name, (h, w, oh, ow, seed) in {'half': (48, 64, 24, 32, 0), 'odd': (37, 53, 19, 20, 1), 'frac': (75, 124, 48, 64, 2),
                                   'up': (12, 20, 30, 33, 3), 'same_w': (40, 32, 20, 32, 4)}.items():
    img = image(h, w, seed)
    cases[f'{name}_in'] = img
    cases[f'{name}_out'] = pil_resize(img, oh, ow)
# a 4-level pyramid as the datasets build it (every level from the previous one), KITTI-like aspect
raw = image(94, 310, 5)
lvl = pil_resize(raw, 64, 128)
cases['pyr_in'] = raw
for s in range(4):
    if s:
        lvl = pil_resize(lvl, 64 >> s, 128 >> s)
    cases[f'pyr_{s}'] = lvl
# full KITTI size: 375x1242 -> 192x640; only a checksum + a crop are stored
raw = image(375, 1242, 6)
full = pil_resize(raw, 192, 640)
cases['kitti_seed'] = np.array([6])
cases['kitti_crop'] = full[80:112, 300:364].copy()
cases['kitti_sum'] = np.array([int(full.astype(np.int64).sum()), int((full.astype(np.int64) * (np.arange(full.size).reshape(full.shape) % 251)).sum())])
np.savez_compressed(OUT / 'resize_pillow.npz', pillow_version=np.array(PIL.__version__), **cases)
print('wrote', OUT / 'resize_pillow.npz', {k: v.shape for k, v in cases.items()})
