"""Thin tensor-level wrappers over the C ABI (pointer / shape plumbing only; no math here)."""
import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_ELU, ACT_NONE, ACT_RELU, PAD_REFLECT, PAD_ZERO  # noqa: F401


def _stream(t: torch.Tensor):
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    lib = _lib.get_lib()
    if t.device.type != lib.device_type:
        raise _lib.ClslamError(f'tensor on {t.device}, library {lib.path.name} runs on {lib.device_type}')
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.ClslamError(f'expected contiguous fp32, got {t.dtype} contiguous={t.is_contiguous()}')
    return t.data_ptr()


def conv2d(src_a, weight, out, *, src_b=None, scale=None, shift=None, residual=None, ksize=3, stride=1,
           pad=None, pad_mode=PAD_ZERO, upsample_a=False, act=ACT_NONE, config=-1):
    """src_a (B,Ha,Wa,Ca) NHWC; weight (Cout, k*k, Ca+Cb); out (B,Ho,Wo,Cout)."""
    B, Ho, Wo, Cout = out.shape
    Ha, Wa, Ca = src_a.shape[1:]
    Hi, Wi = (Ha * 2, Wa * 2) if upsample_a else (Ha, Wa)
    Cb = 0 if src_b is None else src_b.shape[3]
    if pad is None:
        pad = ksize // 2
    assert weight.shape[0] == Cout and weight.numel() == Cout * ksize * ksize * (Ca + Cb), weight.shape
    d = _lib.ConvDesc(_p(src_a), _p(src_b), _p(weight), _p(scale), _p(shift), _p(residual), _p(out),
                      B, Hi, Wi, Ca, Cb, Ho, Wo, Cout, ksize, stride, pad, pad_mode, int(upsample_a), act, config)
    _lib.get_lib().call('clslam_conv2d', C.byref(d), _stream(out))
    return out
