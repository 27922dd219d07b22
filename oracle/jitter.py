"""TEST INFRASTRUCTURE: numpy restatement of the colour augmentation of the reference's datasets
(``datasets/utils.py:236-259`` get_random_color_jitter: brightness / contrast /
saturation / hue adjustments of torchvision.transforms.functional applied in a random order to PIL images).

torchvision 0.11.1 is not part of /root/reference (nor installed); for PIL inputs its functional_pil
implementations are thin calls into Pillow, restated here from the published source:
  adjust_brightness = ImageEnhance.Brightness(img).enhance(f)  = Image.blend(black, img, f)
  adjust_contrast   = ImageEnhance.Contrast(img).enhance(f)    = Image.blend(gray(int(mean(L)+0.5)), img, f)
  adjust_saturation = ImageEnhance.Color(img).enhance(f)       = Image.blend(L-of-img as RGB, img, f)
  adjust_hue        = HSV conversion, h += uint8(f*255) with uint8 wrap-around, back to RGB
and Pillow's own C paths (libImaging Blend.c, Convert.c rgb2hsv_row / hsv2rgb / rgb2l):
  blend      : float32 a + alpha*(b-a) (no fused multiply-add); alpha in [0,1] truncates, otherwise clips
  L          : (r*19595 + g*38470 + b*7471 + 0x8000) >> 16
  RGB<->HSV  : colorsys with Pillow's float/double mix (reproduced operation by operation below)

Pinned: tests/test_ingest.py holds every function bit-exactly to outputs of the REAL Pillow
(tests/golden/jitter_pillow.npz, tests/golden/make_jitter_golden.py); the two HSV conversions were also checked
exhaustively over all 2^24 inputs in the build container.
"""
from typing import Sequence

import numpy as np

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


def to_l(img: np.ndarray) -> np.ndarray:
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(a: np.ndarray, b: np.ndarray, alpha: float) -> np.ndarray:
    al = np.float32(alpha)
    t = a.astype(np.float32) + al * (b.astype(np.float32) - a.astype(np.float32))
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.uint8)                                   # (UINT8) cast: truncation
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t)).astype(np.uint8)


def rgb_to_hsv(img: np.ndarray) -> np.ndarray:
    r, g, b = (img[..., k].astype(np.int32) for k in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        s = cr / maxc.astype(np.float32)
        rc, gc, bc = ((maxc - c).astype(np.float32) / cr for c in (r, g, b))
        f64 = np.float64
        h = np.where(r == maxc, (bc - gc).astype(np.float32),
                     np.where(g == maxc, (2.0 + rc.astype(f64) - bc.astype(f64)).astype(np.float32),
                              (4.0 + gc.astype(f64) - rc.astype(f64)).astype(np.float32)))
        h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(np.float32)
        uh = np.clip((h.astype(f64) * 255.0).astype(np.int64), 0, 255)
        us = np.clip((s.astype(f64) * 255.0).astype(np.int64), 0, 255)
    gray = minc == maxc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv_to_rgb(hsv: np.ndarray) -> np.ndarray:
    f64 = np.float64
    h = hsv[..., 0].astype(np.float32)
    v = hsv[..., 2].astype(np.float32).astype(f64)
    h6 = h.astype(f64) * 6.0 / 255.0
    i = np.floor(h6).astype(np.int64)
    f = (h6 - i.astype(f64)).astype(np.float32).astype(f64)
    fs = (hsv[..., 1].astype(np.float32).astype(f64) / 255.0).astype(np.float32).astype(f64)

    def c_round(x):  # C round(): halves away from zero
        return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)
    p, q, t = (np.clip(c_round(v * e), 0, 255) for e in (1.0 - fs, 1.0 - fs * f, 1.0 - fs * (1.0 - f)))
    vv = hsv[..., 2].astype(np.int64)
    sel = i % 6
    out = np.stack([np.choose(sel, [vv, q, p, p, t, vv]), np.choose(sel, [t, vv, vv, q, p, p]),
                    np.choose(sel, [p, p, t, vv, vv, q])], -1)
    return np.where((hsv[..., 1] == 0)[..., None], vv[..., None], out).astype(np.uint8)


def hue_shift(factor: float) -> int:
    """np.uint8(hue_factor * 255) of functional_pil.adjust_hue: C truncation, then wrap to uint8."""
    return int(factor * 255) % 256


def adjust(img: np.ndarray, op: int, factor: float) -> np.ndarray:
    if op == BRIGHTNESS:
        return blend(np.zeros_like(img), img, factor)
    if op == CONTRAST:
        lum = to_l(img)
        mean = int(int(lum.astype(np.int64).sum()) / lum.size + 0.5)      # ImageStat mean = sum / count in Python
        return blend(np.full_like(img, mean), img, factor)
    if op == SATURATION:
        return blend(np.repeat(to_l(img)[..., None], 3, -1), img, factor)
    if op == HUE:
        hsv = rgb_to_hsv(img)
        hsv[..., 0] = (hsv[..., 0].astype(np.int64) + hue_shift(factor)) % 256
        return hsv_to_rgb(hsv)
    raise ValueError(op)


def color_jitter(img: np.ndarray, order: Sequence[int], factors: Sequence[float]) -> np.ndarray:
    """img (H,W,3) uint8; order = op ids in application order; factors indexed by op id."""
    for op in order:
        img = adjust(img, op, factors[op])
    return img
