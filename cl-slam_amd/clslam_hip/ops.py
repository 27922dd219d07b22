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
           pad=None, pad_mode=PAD_ZERO, upsample_a=False, act=ACT_NONE, config=-1, actgrad_src=None,
           actgrad_kind=ACT_NONE):
    """src_a (B,Ha,Wa,Ca) NHWC; weight (Cout, k*k, Ca+Cb); out (B,Ho,Wo,Cout)."""
    B, Ho, Wo, Cout = out.shape
    Ha, Wa, Ca = src_a.shape[1:]
    Hi, Wi = (Ha * 2, Wa * 2) if upsample_a else (Ha, Wa)
    Cb = 0 if src_b is None else src_b.shape[3]
    if pad is None:
        pad = ksize // 2
    assert weight.shape[0] == Cout and weight.numel() == Cout * ksize * ksize * (Ca + Cb), weight.shape
    d = _lib.ConvDesc(_p(src_a), _p(src_b), _p(weight), _p(scale), _p(shift), _p(residual), _p(out),
                      B, Hi, Wi, Ca, Cb, Ho, Wo, Cout, ksize, stride, pad, pad_mode, int(upsample_a), act, config,
                      _p(actgrad_src), actgrad_kind)
    _lib.get_lib().call('clslam_conv2d', C.byref(d), _stream(out))
    return out


def conv_desc(src_a, out_shape, *, src_b=None, ksize=3, stride=1, pad=None, pad_mode=PAD_ZERO, upsample_a=False):
    """Geometry-only descriptor of a forward conv (used by wgrad)."""
    B, Ho, Wo, Cout = out_shape
    Ha, Wa, Ca = src_a.shape[1:]
    Hi, Wi = (Ha * 2, Wa * 2) if upsample_a else (Ha, Wa)
    Cb = 0 if src_b is None else src_b.shape[3]
    if pad is None:
        pad = ksize // 2
    return _lib.ConvDesc(_p(src_a), _p(src_b), None, None, None, None, None, B, Hi, Wi, Ca, Cb, Ho, Wo, Cout,
                         ksize, stride, pad, pad_mode, int(upsample_a), ACT_NONE, -1, None, ACT_NONE)


def weight_transpose(w, wt, ch_in_sel=None):
    """w (Cout,taps,Cin) -> wt (Cin_sel,taps,Cout), taps flipped."""
    Cout, taps, Cin = w.shape
    sel = Cin if ch_in_sel is None else ch_in_sel
    assert wt.numel() >= sel * taps * Cout
    _lib.get_lib().call('clslam_weight_transpose', _p(w), _p(wt), Cout, taps, Cin, sel, _stream(w))
    return wt


def fold_act_grad(dxp, yout, dz, *, h, w, ch, border, pool, act):
    B = dxp.shape[0]
    _lib.get_lib().call('clslam_fold_act_grad', _p(dxp), _p(yout), _p(dz), B, h, w, ch, dxp.shape[3], border,
                        int(pool), act, _stream(dz))
    return dz


def wgrad_splits(desc, target_blocks=1024) -> int:
    return _lib.get_lib().cdll.clslam_wgrad_splits(C.byref(desc), target_blocks)


def conv_wgrad(desc, dz, partial, splits):
    _lib.get_lib().call('clslam_conv_wgrad', C.byref(desc), _p(dz), _p(partial), splits, _stream(dz))
    return partial


def reduce_partials(partial, out, n, splits, scale=1.0):
    _lib.get_lib().call('clslam_reduce_partials', _p(partial), _p(out), n, splits, scale, _stream(out))
    return out


def colsum_blocks(rows: int) -> int:
    return _lib.get_lib().cdll.clslam_colsum_blocks(rows)


def colsum(x, partial, rows, ch):
    _lib.get_lib().call('clslam_colsum', _p(x), _p(partial), rows, ch, _stream(x))
    return partial
