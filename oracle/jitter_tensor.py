"""TEST INFRASTRUCTURE: torch restatement of the colour jitter the REPLAY BUFFER applies on every frame of the adaptation path.

``slam/slam.py:98`` builds the replay buffer with ``do_augmentation=True``; ``slam/replay_buffer.py:263-291`` (``_get``) then
draws ONE ``get_random_color_jitter((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-.1, .1))`` per replayed sample
(``datasets/utils.py:236-259``: four ``random.uniform`` draws -- brightness, contrast, saturation, hue -- then ``random.shuffle`` of
the four transforms) and applies it to every ``ToTensor``-ed pyramid level, i.e. to FLOAT TENSORS ``(1, 3, h, w)`` in [0, 1]:
torchvision's tensor code path (``transforms/functional_tensor.py``), not the PIL one the pre-training datasets use
(``oracle/jitter.py``).

PARITY UNPINNED: torchvision 0.11.1 (requirements.txt:6) is neither part of /root/reference nor installed, and the reference
holds no test or golden vector for this step.  The functions below restate the published 0.11.1 ``functional_tensor.py`` operation
by operation (each cites the function it follows); tests/golden/make_replay_jitter_golden.py pins them against the real
torchvision the moment one is importable.  Only tests/ and tests/ref_stubs.py import this module.
"""
import random
from typing import List, Sequence, Tuple

import torch
from torch import Tensor

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


def _blend(img1: Tensor, img2: Tensor, ratio: float) -> Tensor:
    """functional_tensor._blend: (ratio * img1 + (1.0 - ratio) * img2).clamp(0, bound).to(img1.dtype), bound = 1.0 for floats"""
    ratio = float(ratio)
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, 1.0).to(img1.dtype)


def rgb_to_grayscale(img: Tensor) -> Tensor:
    """functional_tensor.rgb_to_grayscale (num_output_channels = 1): 0.2989 r + 0.587 g + 0.114 b, keepdim channel"""
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(dim=-3)


def adjust_brightness(img: Tensor, f: float) -> Tensor:
    return _blend(img, torch.zeros_like(img), f)


def adjust_contrast(img: Tensor, f: float) -> Tensor:
    mean = torch.mean(rgb_to_grayscale(img).to(img.dtype), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, f)


def adjust_saturation(img: Tensor, f: float) -> Tensor:
    return _blend(img, rgb_to_grayscale(img), f)


def _rgb2hsv(img: Tensor) -> Tensor:
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc = (maxc - r) / cr_divisor
    gc = (maxc - g) / cr_divisor
    bc = (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = hr + hg + hb
    h = torch.fmod((h / 6.0 + 1.0), 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img: Tensor) -> Tensor:
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp((v * (1.0 - s)), 0.0, 1.0)
    q = torch.clamp((v * (1.0 - f * s)), 0.0, 1.0)
    t = torch.clamp((v * (1.0 - (s * (1.0 - f)))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6, device=i.device).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum('...ijk, ...xijk -> ...xjk', mask.to(dtype=img.dtype), a4)


def adjust_hue(img: Tensor, hue_factor: float) -> Tensor:
    """functional_tensor.adjust_hue for a float image: HSV round trip with h = (h + hue_factor) % 1.0"""
    if not (-0.5 <= hue_factor <= 0.5):
        raise ValueError(f'hue_factor ({hue_factor}) is not in [-0.5, 0.5].')
    hsv = _rgb2hsv(img)
    h, s, v = hsv.unbind(dim=-3)
    h = (h + hue_factor) % 1.0
    return _hsv2rgb(torch.stack((h, s, v), dim=-3))


OPS = {BRIGHTNESS: adjust_brightness, CONTRAST: adjust_contrast, SATURATION: adjust_saturation, HUE: adjust_hue}


def draw_jitter(brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=(-.1, .1), rng=random) -> Tuple[List[int], List[float]]:
    """The draws of datasets/utils.py:236-259 in their order: four uniforms, then the shuffle of the transform list.
    -> (op ids in application order, factors indexed by op id)."""
    factors = [rng.uniform(*brightness), rng.uniform(*contrast), rng.uniform(*saturation), rng.uniform(*hue)]
    order = [BRIGHTNESS, CONTRAST, SATURATION, HUE]
    rng.shuffle(order)
    return order, factors


def color_jitter(img: Tensor, order: Sequence[int], factors: Sequence[float]) -> Tensor:
    """img (..., 3, h, w) float in [0, 1]; every leading index is its own image for the contrast mean only when the caller passes
    one image at a time (the reference does: (1, 3, h, w) per level, replay_buffer.py:277-283)."""
    for op in order:
        img = OPS[op](img, factors[op])
    return img
