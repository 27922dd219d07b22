"""Deterministic synthetic weights and sample dicts (no dataset / checkpoint needed).

Used by bench.py, the tests and the golden generator so that the reference (in the build
container), the CPU oracle and the HIP path all see bit-identical weights and inputs.
Values come from an integer hash (exact in fp32 on every platform), images from a few 2-D
sinusoids so that the view-synthesis warp has texture to work with.

The sample dict follows the reference's input contract (SURVEY.md 8a row A0:
datasets/kitti.py:231-317, datasets/utils.py:104-110,154-171).
"""
import math
from typing import Any, Dict

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def hash_uniform(n: int, seed: int) -> np.ndarray:
    """n floats in [0,1), exact multiples of 2^-24, from a splitmix64-style integer hash."""
    with np.errstate(over='ignore'):
        x = np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def _key_seed(key: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in key.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x632BE59BD9B4E019)) & 0x7FFFFFFFFFFFFFFF


def fill_state_dict(sd: Dict[str, torch.Tensor], seed: int = 0, model: str = '') -> Dict[str, torch.Tensor]:
    """Return a new state dict with the same keys/shapes filled with closed-form values.
    Scales are chosen so activations stay O(1) through the ResNet and the decoder outputs are
    not saturated (checked in tests/test_synth.py)."""
    out = {}
    for k, v in sd.items():
        n = v.numel()
        u = torch.from_numpy(hash_uniform(max(n, 1), _key_seed(model + '/' + k, seed))[:n]).reshape(v.shape)
        if k.endswith('num_batches_tracked'):
            out[k] = v.clone()
            continue
        if k in ('height', 'width'):
            out[k] = v.clone()
            continue
        if v.dim() == 4:  # conv weight (O,I,kh,kw)
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            gain = 1.0
            if 'dispconv' in k:
                gain = 0.5
            elif 'pose_2' in k:
                gain = 2.0
            a = gain * math.sqrt(6.0 / fan_in)
            out[k] = ((u * 2 - 1) * a).to(v.dtype)
        elif k.endswith('running_var'):
            out[k] = (0.6 + 0.8 * u).to(v.dtype)
        elif k.endswith('running_mean'):
            out[k] = (0.2 * (u - 0.5)).to(v.dtype)
        elif '.bn' in k or 'downsample.1' in k:
            if k.endswith('weight'):
                # damp the residual branch (bn2) so 8 blocks do not blow activations up
                base = 0.35 if 'bn2' in k else 0.9
                out[k] = (base + 0.2 * u).to(v.dtype)
            else:
                out[k] = (0.1 * (u - 0.5)).to(v.dtype)
        elif v.dim() == 2:  # fc weight (unused ImageNet head)
            out[k] = ((u * 2 - 1) * 0.01).to(v.dtype)
        else:  # conv / fc bias
            out[k] = (0.1 * (u - 0.5)).to(v.dtype)
    return out


def camera_matrices(H: int, W: int, scale: int = 0):
    """KITTI normalised intrinsics (datasets/kitti.py:65-66) scaled to the pyramid level,
    inverse via pinv (datasets/utils.py:104-110)."""
    K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    K[0, :] *= W // (2**scale)
    K[1, :] *= H // (2**scale)
    return K, np.linalg.pinv(K)


def _texture(H: int, W: int, seed: int, shift: float) -> np.ndarray:
    """3xHxW image in [0,1]: 8 sinusoids per channel + 0.05 hash noise, horizontally shifted."""
    u = hash_uniform(8 * 3 * 4, seed).reshape(3, 8, 4).astype(np.float64)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    xs = xs + shift
    img = np.zeros((3, H, W))
    for c in range(3):
        for j in range(8):
            fx = (0.5 + 7.5 * u[c, j, 0]) * 2 * math.pi / 128.0
            fy = (0.5 + 5.5 * u[c, j, 1]) * 2 * math.pi / 64.0
            ph = 2 * math.pi * u[c, j, 2]
            amp = 0.25 + 0.75 * u[c, j, 3]
            img[c] += amp * np.sin(fx * xs + fy * ys + ph)
        img[c] = 0.5 + img[c] / 8.0
    noise = hash_uniform(3 * H * W, seed + 7919).reshape(3, H, W)
    return np.clip(img + 0.05 * (noise - 0.5), 0.0, 1.0).astype(np.float32)


def _down2(img: np.ndarray) -> np.ndarray:
    C, H, W = img.shape
    return img.reshape(C, H // 2, 2, W // 2, 2).mean(axis=(2, 4)).astype(np.float32)


def make_batch(B: int, H: int, W: int, seed: int = 0, num_scales: int = 4) -> Dict[Any, torch.Tensor]:
    """Synthetic sample dict for B triplets (sample 0 = "online": rgb_aug == rgb; others are
    brightness/contrast-jittered like replay samples, slam/replay_buffer.py:263-291)."""
    d: Dict[Any, list] = {}
    for b in range(B):
        s = seed * 1000 + b
        u = hash_uniform(8, s + 31)
        shift = 2.0 + 4.0 * float(u[0])
        gain = 1.0 if b == 0 else 0.8 + 0.4 * float(u[1])
        contrast = 1.0 if b == 0 else 0.8 + 0.4 * float(u[2])
        for f in (-1, 0, 1):
            img = _texture(H, W, s, shift * f)
            aug = np.clip((img - 0.5) * contrast + 0.5, 0, 1) * gain
            aug = np.clip(aug, 0, 1).astype(np.float32)
            for sc in range(num_scales):
                d.setdefault(('rgb', f, sc), []).append(img)
                d.setdefault(('rgb_aug', f, sc), []).append(aug)
                img, aug = _down2(img), _down2(aug)
        for sc in range(num_scales):
            K, Kinv = camera_matrices(H, W, sc)
            d.setdefault(('camera_matrix', sc), []).append(K)
            d.setdefault(('inv_camera_matrix', sc), []).append(Kinv.astype(np.float32))
        d.setdefault(('relative_distance', 0), []).append(0.2 + 1.3 * float(u[3]))
        d.setdefault(('relative_distance', 1), []).append(0.2 + 1.3 * float(u[4]))
    out: Dict[Any, torch.Tensor] = {'index': torch.arange(B, dtype=torch.int64)}
    for k, v in d.items():
        if k[0] == 'relative_distance':
            out[k] = torch.tensor(v, dtype=torch.float64)  # python floats collate to float64
        else:
            out[k] = torch.from_numpy(np.stack(v, 0))
    return out


def make_noise(B: int, H: int, W: int, seed: int, num_scales: int = 4) -> Dict[int, torch.Tensor]:
    """Tie-break noise tensors (dpp.py:1055-1056: randn * 1e-5) as injected in parity runs:
    approximately normal (sum of 4 uniforms), deterministic."""
    out = {}
    for s in range(num_scales):
        u = hash_uniform(4 * B * 2 * H * W, seed * 100 + s + 1).reshape(4, B, 2, H, W)
        z = (u.sum(0) - 2.0) * math.sqrt(3.0)
        out[s] = torch.from_numpy((z * 1e-5).astype(np.float32))
    return out
