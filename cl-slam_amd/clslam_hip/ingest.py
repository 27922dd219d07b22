"""Image-pyramid ingest on the GPU: uint8 frames uploaded once -> the ``('rgb', f, s)``
float pyramids of the sample dict, bit-identical to what the reference's datasets build with PIL
(``datasets/utils.py:62-66,154-163,213-215``: LANCZOS resize of every level from the previous one, ToTensor).

    pyr = ImagePyramid(192, 640)                    # model resolution, scales 0..3
    levels = pyr(frames_u8)                         # (N,Hraw,Wraw,3) uint8 on the GPU -> {s: (N,3,H>>s,W>>s) float32}

The tap plans (Pillow's double-precision Lanczos-3 weights in 22-bit fixed point) are computed once per size pair
by the library's host function and kept on the device; the kernels are integer multiply-accumulates.
``color_jitter`` applies the datasets' colour augmentation (``datasets/utils.py:236-259``: torchvision's brightness /
contrast / saturation / hue adjustments of PIL images in a drawn order) with Pillow's arithmetic, also bit-exact.
"""
import ctypes as C
from typing import Dict, Sequence, Tuple

import torch

from . import _lib
from .ops import _p, _pa, _stream


class ImagePyramid:
    def __init__(self, height: int, width: int, scales: Sequence[int] = (0, 1, 2, 3)) -> None:
        self.height, self.width, self.scales = int(height), int(width), tuple(scales)
        self._plans: Dict[Tuple[int, int, str], Tuple[torch.Tensor, torch.Tensor, int]] = {}

    def _plan(self, in_size: int, out_size: int, device: torch.device):
        key = (in_size, out_size, str(device))
        plan = self._plans.get(key)
        if plan is None:
            lib = _lib.get_lib()
            ksize = lib.cdll.clslam_lanczos_ksize(in_size, out_size)
            bounds = torch.empty(out_size, 2, dtype=torch.int32)
            coeffs = torch.empty(out_size, ksize, dtype=torch.int32)
            lib.call('clslam_lanczos_plan', in_size, out_size, bounds.data_ptr(), coeffs.data_ptr())   # host function
            plan = (bounds.to(device), coeffs.to(device), ksize)
            self._plans[key] = plan
        return plan

    def _resize(self, src: torch.Tensor, out_h: int, out_w: int, planar: torch.Tensor) -> torch.Tensor:
        """src (N,h,w,C) uint8 -> (N,out_h,out_w,C) uint8; planar receives ToTensor(result)."""
        N, h, w, Cc = src.shape
        lib = _lib.get_lib()
        cur = src
        if w != out_w:                                        # Pillow: horizontal pass first
            b, k, ks = self._plan(w, out_w, src.device)
            dst = torch.empty(N, h, out_w, Cc, dtype=torch.uint8, device=src.device)
            last = h == out_h
            lib.call('clslam_resize_pass_u8', _pa(cur, torch.uint8), _pa(dst, torch.uint8), _p(planar) if last else None,
                     _pa(b, torch.int32), _pa(k, torch.int32), ks, N, h, w, Cc, out_w, 1, _stream(src))
            cur = dst
        if h != out_h:
            b, k, ks = self._plan(h, out_h, src.device)
            dst = torch.empty(N, out_h, out_w, Cc, dtype=torch.uint8, device=src.device)
            lib.call('clslam_resize_pass_u8', _pa(cur, torch.uint8), _pa(dst, torch.uint8), _p(planar), _pa(b, torch.int32),
                     _pa(k, torch.int32), ks, N, h, out_w, Cc, out_h, 0, _stream(src))
            cur = dst
        if cur is src:                                        # already at the target size: ToTensor only
            lib.call('clslam_u8_to_planar_f32', _pa(src, torch.uint8), _p(planar), N, h, w, Cc, _stream(src))
        return cur

    def __call__(self, frames: torch.Tensor, return_u8: bool = False):
        """-> {scale: (N,C,h,w) float32}; with return_u8 also {scale: (N,h,w,C) uint8} (the reference jitters every
        level's uint8 image separately, kitti.py:345-347)."""
        if frames.dim() == 3:
            frames = frames.unsqueeze(0)
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] not in (1, 3, 4):
            raise _lib.ClslamError(f'ImagePyramid expects (N,H,W,C) uint8 frames, got {tuple(frames.shape)} {frames.dtype}')
        frames = frames.contiguous()
        N, Cc = frames.shape[0], frames.shape[-1]
        out: Dict[int, torch.Tensor] = {}
        out_u8: Dict[int, torch.Tensor] = {}
        cur = frames
        for s in range(max(self.scales) + 1):
            h, w = self.height >> s, self.width >> s
            planar = torch.empty(N, Cc, h, w, device=frames.device)
            cur = self._resize(cur, h, w, planar)
            if s in self.scales:
                out[s] = planar
                out_u8[s] = cur
        return (out, out_u8) if return_u8 else out


def to_tensor(frames: torch.Tensor) -> torch.Tensor:
    """ToTensor (datasets/utils.py:213-215) of (N,H,W,C) uint8 frames -> (N,C,H,W) float32 = byte / 255."""
    frames = frames.contiguous()
    N, H, W, Cc = frames.shape
    planar = torch.empty(N, Cc, H, W, device=frames.device)
    _lib.get_lib().call('clslam_u8_to_planar_f32', _pa(frames, torch.uint8), _p(planar), N, H, W, Cc, _stream(frames))
    return planar


BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


def color_jitter(frames: torch.Tensor, order: Sequence[int], factors: Sequence[float]) -> torch.Tensor:
    """frames (N,H,W,3) uint8 on the library's device -> jittered copy.  ``order``: op ids in application order (as
    ``random.shuffle`` left them in get_random_color_jitter), ``factors`` = (brightness, contrast, saturation, hue)."""
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise _lib.ClslamError(f'color_jitter expects (N,H,W,3) uint8 frames, got {tuple(frames.shape)} {frames.dtype}')
    frames = frames.contiguous()
    N, H, W, _ = frames.shape
    out, scratch = torch.empty_like(frames), torch.empty_like(frames)
    lsum = torch.zeros(N, dtype=torch.int64, device=frames.device)
    order_c = (C.c_int * max(len(order), 1))(*[int(o) for o in order])
    factors_c = (C.c_double * 4)(*[float(f) for f in factors])
    _lib.get_lib().call('clslam_color_jitter_u8', _pa(frames, torch.uint8), _pa(out, torch.uint8), _pa(scratch, torch.uint8),
                        _pa(lsum, torch.int64), N, H, W, order_c, len(order), factors_c, _stream(frames))
    return out
