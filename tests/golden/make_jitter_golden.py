#!/usr/bin/env python
"""Generates tests/golden/jitter_pillow.npz with the REAL Pillow of the build container.

The reference augments PIL images with torchvision.transforms.functional.adjust_{brightness,contrast,
saturation,hue} in a random order (datasets/utils.py:236-259).  torchvision is not installed here; for PIL
inputs those four functions are these calls into Pillow (torchvision 0.11.1, functional_pil.py), which this script
makes directly:
    brightness/contrast/saturation: ImageEnhance.{Brightness,Contrast,Color}(img).enhance(factor)
    hue: h,s,v = img.convert('HSV').split(); h += np.uint8(factor*255) (uint8 wrap); merge; convert('RGB')

    python tests/golden/make_jitter_golden.py
"""
from pathlib import Path

import numpy as np
import PIL
from PIL import Image, ImageEnhance

OUT = Path(__file__).resolve().parent


def image(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    ramp = (((x * 5 + y * 3) % 256)[..., None] * np.array([1, 0.5, 0.25])).astype(np.int64)
    noise = rng.integers(0, 256, (h, w, 3))
    img = ((ramp + noise) // 2 % 256).astype(np.uint8)
    img[:4, :8] = 0
    img[4:8, :8] = 255
    img[8:12, :8] = 77           # grey pixels: the s == 0 branches of the HSV conversions
    return img


def pil_adjust(img, op, f):
    if op == 0:
        return ImageEnhance.Brightness(img).enhance(f)
    if op == 1:
        return ImageEnhance.Contrast(img).enhance(f)
    if op == 2:
        return ImageEnhance.Color(img).enhance(f)
    h, s, v = img.convert('HSV').split()
    np_h = np.array(h, dtype=np.uint8)
    np_h = (np_h.astype(np.int64) + int(f * 255) % 256).astype(np.uint8)    # == np_h += np.uint8(f * 255) with wrap
    return Image.merge('HSV', (Image.fromarray(np_h, 'L'), s, v)).convert('RGB')


cases = {}
img = image(40, 56, 11)
cases['in'] = img
pil = Image.fromarray(img)
cases['hsv'] = np.asarray(pil.convert('HSV'))
cases['l'] = np.asarray(pil.convert('L'))
for op, name in enumerate(('brightness', 'contrast', 'saturation')):
    for f in (0.8, 0.93, 1.0, 1.07, 1.2):
        cases[f'{name}_{f}'] = np.asarray(pil_adjust(pil, op, f))
for f in (-0.1, -0.037, 0.0, 0.05, 0.1):
    cases[f'hue_{f}'] = np.asarray(pil_adjust(pil, 3, f))
# full jitters: (order, factors[brightness, contrast, saturation, hue])
JITTERS = [((0, 1, 2, 3), (0.85, 1.15, 0.9, 0.08)), ((3, 2, 1, 0), (1.19, 0.81, 1.2, -0.1)), ((2, 0, 3, 1), (1.0, 1.1, 0.8, -0.04)),
           ((1, 3), (1.0, 0.9, 1.0, 0.02))]
for n, (order, factors) in enumerate(JITTERS):
    cur = pil
    for op in order:
        cur = pil_adjust(cur, op, factors[op])
    cases[f'jitter{n}_order'] = np.array(order)
    cases[f'jitter{n}_factors'] = np.array(factors, dtype=np.float64)
    cases[f'jitter{n}_out'] = np.asarray(cur)
np.savez_compressed(OUT / 'jitter_pillow.npz', pillow_version=np.array(PIL.__version__), **cases)
print('wrote', OUT / 'jitter_pillow.npz', len(cases), 'arrays')
