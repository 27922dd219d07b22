"""Thin tensor-level wrappers over the C ABI (pointer / shape plumbing only; no math here)."""
import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_ELU, ACT_HSIGMOID, ACT_HSWISH, ACT_NONE, ACT_RELU, PAD_REFLECT, PAD_ZERO  # noqa: F401


# profile_begin() sets this to a list collecting (kind, tile config, algorithmic flops, description) per conv launch
PROFILE = None


def profile_begin(max_launches: int = 8192) -> None:
    """Measurement (bench.py's roofline leg): every conv2d() launch from now on is timestamped by the library itself
    (clslam_conv_profile_begin: kernel start/stop, the duration rocprofv3 reports).  Run the launches in serial order
    (engine.use_side_stream = False): concurrent kernels share the GPU and a launch's duration says little then."""
    global PROFILE
    _lib.get_lib().call('clslam_conv_profile_begin', max_launches)
    PROFILE = []


def profile_end():
    """-> [(kind, config, flops, seconds, description, algorithmic_bytes)] in launch order; disarms."""
    global PROFILE
    meta, PROFILE = PROFILE, None
    cap = max(1, len(meta))
    ms = (C.c_float * cap)()
    count = C.c_int(0)
    _lib.get_lib().call('clslam_conv_profile_end', C.cast(ms, C.c_void_p), cap, C.cast(C.pointer(count), C.c_void_p))
    if not _lib.get_lib().is_device:      # the emulator build has no timestamps
        return [(k, cfg, fl, 0.0, desc, nb) for k, cfg, fl, desc, nb in meta]
    if count.value != len(meta):
        raise _lib.ClslamError(f'profile_end: {count.value} timed launches for {len(meta)} conv2d calls')
    return [(k, cfg, fl, ms[i] * 1e-3, desc, nb) for i, (k, cfg, fl, desc, nb) in enumerate(meta)]


# These two helpers run ~1000 times per step; at B <= 2 the step is host-bound, so they avoid every avoidable
# Python-level call (torch.cuda.current_stream() builds a Stream object: ~2 us each, 200 times per step).
_raw_stream = torch._C._cuda_getCurrentRawStream if hasattr(torch._C, '_cuda_getCurrentRawStream') else None


_FORCED_STREAM: Optional[int] = None


class launch_on:
    """`with launch_on(stream):` -- every launch of this module inside the block goes to `stream` (a torch.cuda.Stream).
    Unlike `torch.cuda.stream()` (6 us per block, ~25 blocks per step: the B = 1 step is bound by the host) it does not touch
    torch's current stream: only for blocks that contain nothing but launches of this module."""
    __slots__ = ('handle', 'prev')

    def __init__(self, stream):
        self.handle = None if stream is None else stream.cuda_stream

    def __enter__(self):
        global _FORCED_STREAM
        self.prev = _FORCED_STREAM
        if self.handle is not None:
            _FORCED_STREAM = self.handle
        return self

    def __exit__(self, *exc):
        global _FORCED_STREAM
        _FORCED_STREAM = self.prev
        return False


def _stream(t: torch.Tensor) -> int:
    """raw hipStream_t the launch goes to: the launch_on() stream if one is set, else torch's current stream on t's device
    (0 = the emulator's only stream)"""
    if _FORCED_STREAM is not None:
        return _FORCED_STREAM
    if t.is_cuda:
        if _raw_stream is not None:
            return _raw_stream(t.device.index)
        return torch.cuda.current_stream(t.device).cuda_stream
    return 0


_QUERY_CACHE: dict = {}


def _query(name: str, *args) -> int:
    """Launch-geometry queries of the library (pure functions of integer arguments): asked once, then remembered -- a bare
    ctypes call costs ~10 us and fold_blocks() alone was asked ten times per step."""
    lib = _lib._LIB or _lib.get_lib()
    key = (id(lib), name, args)
    v = _QUERY_CACHE.get(key)
    if v is None:
        v = _QUERY_CACHE[key] = getattr(lib.cdll, name)(*args)
    return v


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    lib = _lib._LIB or _lib.get_lib()
    if t.is_cuda != lib.is_device or t.dtype is not torch.float32 or not t.is_contiguous():
        if t.device.type != lib.device_type:
            raise _lib.ClslamError(f'tensor on {t.device}, library {lib.path.name} runs on {lib.device_type}')
        raise _lib.ClslamError(f'expected contiguous fp32, got {t.dtype} contiguous={t.is_contiguous()}')
    return t.data_ptr()


def _pa(t: Optional[torch.Tensor], dtype):
    """pointer of a contiguous tensor of another dtype (uint8 selection maps, float64 distances)"""
    if t is None:
        return None
    lib = _lib.get_lib()
    if t.device.type != lib.device_type or t.dtype != dtype or not t.is_contiguous():
        raise _lib.ClslamError(f'expected contiguous {dtype} on {lib.device_type}, got {t.dtype} on {t.device}')
    return t.data_ptr()


class Handoff:
    """Releases a consumer stream behind ONE launch of a producer stream without a marker packet on the producer
    (clslam_handoff_*, include/clslam_hip.h): `arm()` right before the producing conv2d() / fold_act_grad() call, `release(producer,
    consumer)` after it.  A pool of events, re-used round-robin (an event may be re-armed once it has been waited on; the pool is
    much larger than the hand-offs in flight).  On the emulator (one stream) both calls are no-ops."""
    POOL = 64

    def __init__(self) -> None:
        self.lib = _lib.get_lib()
        self.events = []
        self.next = 0
        self.armed = None

    def arm(self) -> None:
        if not self.lib.is_device:
            return
        if len(self.events) < self.POOL:
            ev = self.lib.cdll.clslam_handoff_event_create()
            if not ev:
                raise _lib.ClslamError('clslam_handoff_event_create failed: ' + self.lib.cdll.clslam_last_error().decode())
            self.events.append(ev)
            self.armed = ev
        else:
            self.armed = self.events[self.next % self.POOL]
        self.next += 1
        self.lib.call('clslam_handoff_arm', self.armed)

    def release(self, producer, consumer) -> None:
        """`consumer` (torch.cuda.Stream) waits for the armed launch -- or, had none taken the event, for everything enqueued on
        `producer` so far"""
        ev, self.armed = self.armed, None
        if ev is None:
            return
        self.lib.call('clslam_handoff_wait', ev, producer.cuda_stream, consumer.cuda_stream)

    def __del__(self):
        try:
            for ev in self.events:
                self.lib.cdll.clslam_handoff_event_destroy(ev)
        except Exception:
            pass


# split-K scratch per stream (stream handle -> zero-initialised uint8 tensor), see clslam_conv_desc.workspace
_CONV_WORKSPACES = {}


def set_conv_workspace(stream_handle: int, workspace: Optional[torch.Tensor]) -> None:
    """Register the split-K scratch conv2d() uses for launches on `stream_handle` (None: drop it).  The
    tensor must be zero-filled once and used by that stream only."""
    if workspace is None:
        _CONV_WORKSPACES.pop(stream_handle, None)
    else:
        assert workspace.dtype == torch.uint8 and workspace.is_contiguous()
        _CONV_WORKSPACES[stream_handle] = workspace


# clslam_conv_desc.cu_limit of the conv launches that follow (0 = the whole chip): set by the engine per step, see there
PERSISTENT_CU_LIMIT = 0


# Descriptors of the engine's fixed call sites (conv2d(..., key=...)): the same layer on the same workspace buffers every step.
# Building the 28-field ctypes structure costs 2.2 of the 5.8 us a conv2d() call takes on the MI355X box's host; a single-triplet
# step (75 convolutions, host-bound) pays that 75 times.  An entry is re-used only while the three tensors that can change under
# a call site (source, weight, output) still sit where they sat; everything else of a site is fixed by construction.
_CONV_DESC_CACHE: dict = {}
_DESC_CACHE_ON = __import__('os').environ.get('CLSLAM_DESC_CACHE', '1') != '0'


def conv2d(src_a, weight, out, *, src_b=None, scale=None, shift=None, residual=None, ksize=3, stride=1,
           pad=None, pad_mode=PAD_ZERO, upsample_a=False, act=ACT_NONE, config=-1, actgrad_src=None,
           actgrad_kind=ACT_NONE, workspace=None, weight_wino=None, cu_limit=None, key=None, cache=None):
    """src_a (B,Ha,Wa,Ca) NHWC; weight (Cout, k*k, Ca+Cb); out (B,Ho,Wo,Cout); weight_wino: wino_weight_transform(weight).
    key (hashable, optional): identifies a FIXED call site -- same layer, same buffers, same keyword arguments every time;
    its descriptor is kept and only stream-dependent fields (split-K scratch, cu_limit) are refreshed.
    cache: the dict the descriptor lives in.  The engine passes its OWN (Engine._desc_cache): the entries then die with the
    buffers they point into -- a module-global cache keyed by id() could hand a later engine, whose workspace happens to be
    allocated at the same addresses, a descriptor with another geometry; the global one serves tests and tools."""
    if cache is None:
        cache = _CONV_DESC_CACHE
    if key is not None and PROFILE is None and _DESC_CACHE_ON:
        ent = cache.get(key)
        if ent is not None:
            d, ref, lib = ent
            if d.src_a == src_a.data_ptr() and d.out == out.data_ptr() and d.weight == weight.data_ptr():
                stream = _stream(out)
                ws = workspace if workspace is not None else (_CONV_WORKSPACES.get(stream) if _CONV_WORKSPACES else None)
                if ws is None:
                    d.workspace, d.workspace_bytes = None, 0
                else:
                    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
                d.cu_limit = PERSISTENT_CU_LIMIT if cu_limit is None else cu_limit
                rc = lib.cdll.clslam_conv2d(ref, stream)
                if rc != _lib.OK:
                    raise _lib.ClslamError(f'clslam_conv2d failed ({rc}): {lib.cdll.clslam_last_error().decode()}')
                return out
    B, Ho, Wo, Cout = out.shape
    Ha, Wa, Ca = src_a.shape[1:]
    Hi, Wi = (Ha * 2, Wa * 2) if upsample_a else (Ha, Wa)
    Cb = 0 if src_b is None else src_b.shape[3]
    if pad is None:
        pad = ksize // 2
    assert weight.shape[0] == Cout and weight.numel() == Cout * ksize * ksize * (Ca + Cb), weight.shape
    stream = _stream(out)
    if workspace is None and _CONV_WORKSPACES:
        workspace = _CONV_WORKSPACES.get(stream)
    d = _lib.ConvDesc(_p(src_a), _p(src_b), _p(weight), _p(scale), _p(shift), _p(residual), _p(out),
                      B, Hi, Wi, Ca, Cb, Ho, Wo, Cout, ksize, stride, pad, pad_mode, int(upsample_a), act, config,
                      _p(actgrad_src), actgrad_kind, None if workspace is None else workspace.data_ptr(),
                      0 if workspace is None else workspace.numel(), _p(weight_wino),
                      PERSISTENT_CU_LIMIT if cu_limit is None else cu_limit)
    if PROFILE is not None:     # armed by profile_begin(): the library timestamps the launch itself
        cfg = config if config >= 0 else _lib.get_lib().cdll.clslam_conv2d_pick_config(C.byref(d))
        # algorithmic bytes: every operand once (source a as stored, i.e. before the nearest-2x upsampling)
        nbytes = 4.0 * (src_a.numel() + (0 if src_b is None else src_b.numel()) + weight.numel() + out.numel() +
                        (0 if residual is None else residual.numel()) + (0 if actgrad_src is None else actgrad_src.numel()))
        PROFILE.append(('conv', cfg, 2.0 * B * Ho * Wo * Cout * ksize * ksize * (Ca + Cb),
                        f'B{B} {Hi}x{Wi} {Ca}+{Cb}->{Cout} k{ksize} s{stride} pad{pad}', nbytes))
    lib = _lib.get_lib()
    lib.call('clslam_conv2d', C.byref(d), stream)
    if key is not None and PROFILE is None:
        cache[key] = (d, C.byref(d), lib)
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
                         ksize, stride, pad, pad_mode, int(upsample_a), ACT_NONE, -1, None, ACT_NONE, None, 0, None, 0)


def wino_weight_transform(w, out=None):
    """w (Cout, 9, Cin) -> the Winograd F(2x2,3x3) filter image clslam_conv2d takes as `weight_wino` (conv_wino.hip)."""
    Cout, taps, Cin = w.shape
    assert taps == 9
    n = _lib.get_lib().cdll.clslam_wino_weight_size(Cout, Cin)
    if out is None:
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    assert out.numel() >= n
    _lib.get_lib().call('clslam_wino_weight_transform', _p(w), _p(out), Cout, Cin, _stream(w))
    return out


def weight_transpose(w, wt, ch_in_sel=None):
    """w (Cout,taps,Cin) -> wt (Cin_sel,taps,Cout), taps flipped."""
    Cout, taps, Cin = w.shape
    sel = Cin if ch_in_sel is None else ch_in_sel
    assert wt.numel() >= sel * taps * Cout
    _lib.get_lib().call('clslam_weight_transpose', _p(w), _p(wt), Cout, taps, Cin, sel, _stream(w))
    return wt


def transpose_table(items):
    """items: [(w (Cout,taps,Cin), wt, ch_in_sel | None)] -> a host table for weight_transpose_multi (build once: the
    arena views and the wt buffers of an engine never move)"""
    tab = (_lib.TransposeItem * len(items))()
    for rec, (w, wt, sel) in zip(tab, items):
        Cout, taps, Cin = w.shape
        sel = Cin if sel is None else sel
        assert wt.numel() >= sel * taps * Cout
        rec.w, rec.wt, rec.ch_out, rec.taps, rec.ch_in, rec.ch_in_sel = _p(w), _p(wt), Cout, taps, Cin, sel
    return tab


def weight_transpose_multi(table, stream_ref):
    """every weight tensor of `table` flipped / transposed for its dgrad convolution, in one launch"""
    _lib.get_lib().call('clslam_weight_transpose_multi', table, len(table), _stream(stream_ref))


def fold_blocks(batch, h, w, ch, pool) -> int:
    return _query('clslam_fold_blocks', batch, h, w, ch, int(pool))


def fold_act_grad(dxp, yout, dz, *, h, w, ch, border, pool, act, bias_partial=None, disp_dz=None, disp_w=None):
    """disp_dz (B,h,w) + disp_w (9,ch): also add the dispconv head's data gradient (dxp may be None then)."""
    B = dz.shape[0]
    _lib.get_lib().call('clslam_fold_act_grad', _p(dxp), _p(yout), _p(dz), _p(bias_partial), B, h, w, ch,
                        ch if dxp is None else dxp.shape[3], border, int(pool), act, _p(disp_dz), _p(disp_w), _stream(dz))
    return dz


def wgrad_splits(desc, target_blocks=1024) -> int:
    return _lib.get_lib().cdll.clslam_wgrad_splits(C.byref(desc), target_blocks)


def conv_wgrad(desc, dz, partial, splits):
    _lib.get_lib().call('clslam_conv_wgrad', C.byref(desc), _p(dz), _p(partial), splits, _stream(dz))
    return partial


def wgrad_patch_supported(desc) -> bool:
    return bool(_lib.get_lib().cdll.clslam_wgrad_patch_supported(C.byref(desc)))


def wgrad_patch_splits(desc, target_blocks=1024) -> int:
    return _lib.get_lib().cdll.clslam_wgrad_patch_splits(C.byref(desc), target_blocks)


def conv_wgrad_patch(desc, dz, partial, splits):
    _lib.get_lib().call('clslam_conv_wgrad_patch', C.byref(desc), _p(dz), _p(partial), splits, _stream(dz))
    return partial


def reduce_partials(partial, out, n, splits, scale=1.0):
    _lib.get_lib().call('clslam_reduce_partials', _p(partial), _p(out), n, splits, scale, _stream(out))
    return out


def make_reduce_table(items, device):
    """items: list of (partial tensor, out tensor, n, splits) -> device byte tensor for reduce_multi"""
    import numpy as np
    rec = np.zeros(len(items), dtype=np.dtype([('partial', '<u8'), ('out', '<u8'), ('n', '<u8'), ('splits', '<i4'), ('scale', '<f4')]))
    for i, (part, out, n, splits) in enumerate(items):
        rec[i] = (_p(part), _p(out), n, splits, 1.0)
    return torch.from_numpy(rec.view(np.uint8).copy()).to(device)


def reduce_multi(table, nitems, stream_ref, blocks_per_item=384):
    _lib.get_lib().call('clslam_reduce_multi', table.data_ptr(), nitems, blocks_per_item, _stream(stream_ref))


def reduce_multi_adam(table, nitems, grad, param, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, guard=None,
                      blocks_per_item=384):
    """reduce_multi + the Adam update of every reduced element in the same launch (single-GPU path)"""
    _lib.get_lib().call('clslam_reduce_multi_adam', table.data_ptr(), nitems, blocks_per_item, _p(grad), _p(param), _p(exp_avg),
                        _p(exp_avg_sq), lr, beta1, beta2, eps, step, _p(guard), _stream(grad))


def copy_multi(pairs, stream_ref=None):
    """pairs: [(src, dst), ...] device tensors of equal dtype / byte size, both contiguous -> one launch for all
    copies (instead of one torch copy kernel each)."""
    pairs = [(s, d) for s, d in pairs if d.numel()]
    if not pairs:
        return
    items = (_lib.CopyItem * len(pairs))()
    for it, (src, dst) in zip(items, pairs):
        if (src.dtype != dst.dtype or src.numel() != dst.numel() or src.device != dst.device
                or not src.is_contiguous() or not dst.is_contiguous()):
            raise _lib.ClslamError(f'copy_multi: {tuple(src.shape)} {src.dtype} {src.device} -> '
                                   f'{tuple(dst.shape)} {dst.dtype} {dst.device} is not a plain copy')
        it.src, it.dst, it.bytes = src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size()
    _lib.get_lib().call('clslam_copy_multi', C.cast(items, C.c_void_p), len(pairs), _stream(stream_ref if stream_ref is not None else pairs[0][1]))


def colsum_blocks(rows: int) -> int:
    return _query('clslam_colsum_blocks', rows)


def colsum(x, partial, rows, ch):
    _lib.get_lib().call('clslam_colsum', _p(x), _p(partial), rows, ch, _stream(x))
    return partial


def stem_pack_weight(weight):
    """OIHW (64,3|6,7,7) conv1 weight -> packed LDS-image slabs for stem_conv"""
    n_img = weight.shape[1] // 3
    packed = torch.empty(_lib.get_lib().cdll.clslam_stem_packed_size(n_img), device=weight.device)
    _lib.get_lib().call('clslam_stem_pack_weight', _p(weight), _p(packed), n_img, _stream(weight))
    return packed


def stem_conv(img_a, img_b, weight, scale, shift, out):
    """img_* planar (B,3,H,W); weight = stem_pack_weight(conv1.weight); out NHWC (B,H/2,W/2,64)."""
    B, _, H, W = img_a.shape
    n_img = 1 if img_b is None else 2
    _lib.get_lib().call('clslam_stem_conv', _p(img_a), _p(img_b), _p(weight), _p(scale), _p(shift), _p(out),
                        B, H, W, n_img, _stream(out))
    return out


def maxpool3x3s2(x, out):
    B, H, W, Cc = x.shape
    _lib.get_lib().call('clslam_maxpool3x3s2', _p(x), _p(out), B, H, W, Cc, _stream(out))
    return out


def dispconv_fwd(x, w, bias, disp):
    B, H, W, Cc = x.shape
    _lib.get_lib().call('clslam_dispconv_fwd', _p(x), _p(w), _p(bias), _p(disp), B, H, W, Cc, _stream(disp))
    return disp


def dispconv_bwd_data(dz, w, dxp, ch, accumulate):
    B, H, W = dz.shape[0], dz.shape[-2], dz.shape[-1]
    _lib.get_lib().call('clslam_dispconv_bwd_data', _p(dz), _p(w), _p(dxp), B, H, W, ch, int(accumulate), _stream(dxp))
    return dxp


def dispconv_wgrad_blocks(pixels: int) -> int:
    return _query('clslam_dispconv_wgrad_blocks', pixels)


def dispconv_wgrad(dz, x, partial):
    B, H, W, Cc = x.shape
    _lib.get_lib().call('clslam_dispconv_wgrad', _p(dz), _p(x), _p(partial), B, H, W, Cc, _stream(x))
    return partial


def pose_head_fwd(x, w2, b2, mean, pose):
    N, H, W, _ = x.shape
    _lib.get_lib().call('clslam_pose_head_fwd', _p(x), _p(w2), _p(b2), _p(mean), _p(pose), N, H * W, _stream(x))
    return pose


def pose_head_bwd(dpose, x, w2, mean, dz1, dw2, db2, grad_scale=1.0):
    N, H, W, _ = x.shape
    _lib.get_lib().call('clslam_pose_head_bwd', _p(dpose), _p(x), _p(w2), _p(mean), _p(dz1), _p(dw2), _p(db2),
                        N, H * W, grad_scale, _stream(x))


def _nd(v):
    return -1.0 if v is None else float(v)


def pose_to_proj(pose, kmat, cam_t_cam, proj):
    B = kmat.shape[0]
    _lib.get_lib().call('clslam_pose_to_proj', _p(pose), _p(kmat), _p(cam_t_cam), _p(proj), B, _stream(pose))


def warp_fwd(disp_s, src_m1, src_p1, inv_k, proj, depth, warped, min_depth, max_depth):
    B, H, W = depth.shape[0], depth.shape[-2], depth.shape[-1]
    h, w = disp_s.shape[-2:]
    _lib.get_lib().call('clslam_warp_fwd', _p(disp_s), h, w, _p(src_m1), _p(src_p1), _p(inv_k), _p(proj), _p(depth),
                        _p(warped), B, H, W, _nd(min_depth), _nd(max_depth), _stream(depth))


def warp_bwd_blocks(H, W) -> int:
    return _query('clslam_warp_bwd_blocks', H, W)


def warp_bwd(dpred, disp_s, src_m1, src_p1, inv_k, proj, ddisp_up, dp_partial, min_depth, max_depth):
    B, H, W = ddisp_up.shape[0], ddisp_up.shape[-2], ddisp_up.shape[-1]
    h, w = disp_s.shape[-2:]
    _lib.get_lib().call('clslam_warp_bwd', _p(dpred), _p(disp_s), h, w, _p(src_m1), _p(src_p1), _p(inv_k), _p(proj),
                        _p(ddisp_up), _pa(dp_partial, torch.float64), B, H, W, _nd(min_depth), _nd(max_depth), _stream(dpred))


def pose_bwd(dp_partial, nscale, nblk, pose, kmat, dist0, dist1, sample_w, vel_scale, dpose):
    B = kmat.shape[0]
    _lib.get_lib().call('clslam_pose_bwd', _pa(dp_partial, torch.float64), nscale, nblk, _p(pose), _p(kmat), _pa(dist0, torch.float64),
                        _pa(dist1, torch.float64), _p(sample_w), float(vel_scale or 0.0), _p(dpose), B, _stream(pose))


def photo_map(pred, target, out_map, coef, npred, batch, H, W):
    _lib.get_lib().call('clslam_photo_map', _p(pred), _p(target), _p(out_map), _p(coef), npred, batch, H, W,
                        _stream(out_map))


def automask_blocks(H, W) -> int:
    return _query('clslam_automask_blocks', H, W)


def automask(idmap, noise, rpmap, sel, partial, batch, H, W):
    _lib.get_lib().call('clslam_automask', _p(idmap), _p(noise), _p(rpmap), _pa(sel, torch.uint8), _p(partial), batch,
                        H, W, _stream(idmap))


def disp_mean_chunks() -> int:
    return _query('clslam_disp_mean_chunks')


def disp_mean(disp, means):
    B = disp.shape[0]
    _lib.get_lib().call('clslam_disp_mean', _p(disp), _p(means), B, disp.numel() // B, _stream(disp))


def loss_finalize(partials, disps, rgb0s, means, pose, dist0, dist1, sample_w, smooth_w, losses, smooth_aux, batch,
                  nblk, H, W, n_smooth, smooth_scale, vel_scale):
    d = _lib.LossDesc()
    for s in range(4):
        d.partial[s] = _p(partials[s]); d.disp[s] = _p(disps[s]); d.rgb0[s] = _p(rgb0s[s]); d.means[s] = _p(means[s])
    d.pose = _p(pose)
    d.dist0 = _pa(dist0, torch.float64); d.dist1 = _pa(dist1, torch.float64)
    d.sample_w = _p(sample_w); d.smooth_w = _p(smooth_w); d.losses = _p(losses); d.smooth_aux = _p(smooth_aux)
    d.batch, d.nblk, d.H, d.W, d.n_smooth = batch, nblk, H, W, n_smooth
    d.smooth_scale, d.vel_scale = float(smooth_scale), float(vel_scale or 0.0)
    _lib.get_lib().call('clslam_loss_finalize', C.byref(d), _stream(losses))


def photo_grad(sel, coef, pred, target, sample_w, dpred, batch, H, W):
    _lib.get_lib().call('clslam_photo_grad', _pa(sel, torch.uint8), _p(coef), _p(pred), _p(target), _p(sample_w),
                        _p(dpred), batch, H, W, _stream(dpred))


def disp_grad(ddisp_up, disp, smooth_aux, n_smooth, dz, H, W):
    B, h, w = disp.shape[0], disp.shape[-2], disp.shape[-1]
    _lib.get_lib().call('clslam_disp_grad', _p(ddisp_up), _p(disp), _p(smooth_aux), n_smooth, _p(dz), B, h, w, H, W,
                        _stream(dz))


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, guard=None):
    """guard: optional 1-element fp32 tensor (the step's loss); NaN there turns the launch into a no-op."""
    _lib.get_lib().call('clslam_adam_step', _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, beta1,
                        beta2, eps, step, grad_scale, _p(guard), _stream(param))


# ---- loop-closure encoder (MobileNetV3-small) ops ------------------------------------------------
def mbv3_stem(img, weight, scale, shift, out):
    B, _, H, W = img.shape
    _lib.get_lib().call('clslam_mbv3_stem', _p(img), _p(weight), _p(scale), _p(shift), _p(out), B, H, W, _stream(out))
    return out


def dwconv(x, weight, scale, shift, out, ksize, stride, act):
    B, H, W, Cc = x.shape
    _lib.get_lib().call('clslam_dwconv', _p(x), _p(weight), _p(scale), _p(shift), _p(out), B, H, W, Cc, ksize, stride, act,
                        _stream(out))
    return out


def avgpool_chunks(hw: int) -> int:
    return _query('clslam_avgpool_chunks', hw)


def global_avgpool(x, out, partial=None):
    """partial: optional (B * avgpool_chunks(HW) * C) scratch -> two-stage reduction over pixel chunks."""
    B, Cc = x.shape[0], x.shape[-1]
    hw = 1
    for n in x.shape[1:-1]:
        hw *= n
    _lib.get_lib().call('clslam_global_avgpool', _p(x), _p(out), _p(partial), B, hw, Cc, _stream(out))
    return out


def se_gate(pool, w1, b1, w2, b2, gate):
    B, Cc = pool.shape
    _lib.get_lib().call('clslam_se_gate', _p(pool), _p(w1), _p(b1), _p(w2), _p(b2), _p(gate), B, Cc, w1.shape[0],
                        _stream(gate))
    return gate


def channel_scale(x, gate):
    B, Cc = x.shape[0], x.shape[-1]
    _lib.get_lib().call('clslam_channel_scale', _p(x), _p(gate), B, x.numel() // (B * Cc), Cc, _stream(x))
    return x


# ---- pyramid forms: one launch for the four scales -----------------------------------------------------
def _ptr4(tensors):
    return (C.c_void_p * 4)(*[_p(t) for t in tensors])


def warp_fwd_pyramid(disps, src_m1, src_p1, inv_k, proj, depth, warped, min_depth, max_depth, scales=(0, 4)):
    """scales = (first, count): that part of the pyramid only (same buffers)"""
    B, H, W = depth.shape[1], depth.shape[-2], depth.shape[-1]
    _lib.get_lib().call('clslam_warp_fwd_pyramid_range', _ptr4(disps), _p(src_m1), _p(src_p1), _p(inv_k), _p(proj), _p(depth),
                        _p(warped), B, H, W, _nd(min_depth), _nd(max_depth), scales[0], scales[1], _stream(depth))


def warp_cells_pyramid(disps, inv_k, proj, cells, min_depth, max_depth):
    """diagnostic: cells (4,2,B,H,W) int32 = x0 | y0 << 12 | (x not clipped) << 24 | (y not clipped) << 25"""
    B, H, W = cells.shape[2], cells.shape[-2], cells.shape[-1]
    _lib.get_lib().call('clslam_warp_cells_pyramid', _ptr4(disps), _p(inv_k), _p(proj), _pa(cells, torch.int32), B, H, W,
                        _nd(min_depth), _nd(max_depth), _stream(cells))
    return cells


def warp_coords_pyramid(disps, inv_k, proj, coords, min_depth, max_depth):
    """diagnostic: coords (4,2,B,H,W,2) float32 = the clipped sampling position (ix, iy) in pixels"""
    B, H, W = coords.shape[2], coords.shape[3], coords.shape[4]
    _lib.get_lib().call('clslam_warp_coords_pyramid', _ptr4(disps), _p(inv_k), _p(proj), _p(coords), B, H, W,
                        _nd(min_depth), _nd(max_depth), _stream(coords))
    return coords


def automask_pyramid(idmap, noise, rpmap, sel, partial, batch, H, W):
    _lib.get_lib().call('clslam_automask_pyramid', _p(idmap), _p(noise), _p(rpmap), _pa(sel, torch.uint8), _p(partial), 4,
                        batch, H, W, _stream(idmap))


def disp_mean_pyramid(disps, means, H, W):
    _lib.get_lib().call('clslam_disp_mean_pyramid', _ptr4(disps), _p(means), disps[0].shape[0], H, W, _stream(means))


def loss_bwd_blocks(H, W) -> int:
    return _query('clslam_loss_bwd_blocks', H, W)


def loss_bwd_pyramid(disps, sel, coef, warped, target, src_m1, src_p1, inv_k, proj, sample_w, ddisp_up, dp_partial,
                     min_depth, max_depth):
    B, H, W = ddisp_up.shape[1], ddisp_up.shape[-2], ddisp_up.shape[-1]
    _lib.get_lib().call('clslam_loss_bwd_pyramid', _ptr4(disps), _pa(sel, torch.uint8), _p(coef), _p(warped), _p(target),
                        _p(src_m1), _p(src_p1), _p(inv_k), _p(proj), _p(sample_w), _p(ddisp_up), _pa(dp_partial, torch.float64), B, H, W,
                        _nd(min_depth), _nd(max_depth), _stream(ddisp_up))


def disp_grad_pyramid(ddisp_up, disps, smooth_aux, n_smooth, dzs, H, W, scales=(0, 4)):
    _lib.get_lib().call('clslam_disp_grad_pyramid_range', _p(ddisp_up), _ptr4(disps), _p(smooth_aux), n_smooth, _ptr4(dzs),
                        disps[0].shape[0], H, W, scales[0], scales[1], _stream(ddisp_up))


def photo_automask_pyramid(warped, target, idmap, noise, sel, coef_sel, partial, batch, H, W, scales=(0, 4)):
    _lib.get_lib().call('clslam_photo_automask_pyramid_range', _p(warped), _p(target), _p(idmap), _p(noise), 0, 0, _pa(sel, torch.uint8),
                        _p(coef_sel), _p(partial), batch, H, W, scales[0], scales[1], _stream(warped))


def photo_automask_pyramid_rng(warped, target, idmap, seed, offset, sel, coef_sel, partial, batch, H, W, scales=(0, 4)):
    """tie-break noise drawn in the kernel (Philox): seed != 0, offset = draw counter of the step"""
    if not seed:
        raise _lib.ClslamError('photo_automask_pyramid_rng: seed must be non-zero')
    _lib.get_lib().call('clslam_photo_automask_pyramid_range', _p(warped), _p(target), _p(idmap), None, int(seed), int(offset),
                        _pa(sel, torch.uint8), _p(coef_sel), _p(partial), batch, H, W, scales[0], scales[1], _stream(warped))


def tie_break_noise(out, seed, offset):
    """out (..., 2) fp32: the in-kernel tie-break stream, element e -> out.view(-1, 2)[e]"""
    _lib.get_lib().call('clslam_tie_break_noise', _p(out), out.numel() // 2, int(seed), int(offset), _stream(out))
    return out


def smooth_intended_chunks() -> int:
    return _query('clslam_smooth_intended_chunks')


def smooth_intended_fwd(disps, rgb0, partial, H, W):
    _lib.get_lib().call('clslam_smooth_intended_fwd', _ptr4(disps), _ptr4(rgb0), _p(partial), disps[0].shape[0], H, W,
                        _stream(partial))


def smooth_intended_finalize(partial, means, sample_w, losses, aux, batch, H, W, smooth_scale):
    _lib.get_lib().call('clslam_smooth_intended_finalize', _p(partial), _p(means), _p(sample_w), _p(losses), _p(aux), batch, H, W,
                        float(smooth_scale), _stream(losses))


def smooth_intended_bwd(disps, rgb0, aux, sample_w, dz, H, W, smooth_scale):
    _lib.get_lib().call('clslam_smooth_intended_bwd', _ptr4(disps), _ptr4(rgb0), _p(aux), _p(sample_w), _ptr4(dz),
                        disps[0].shape[0], H, W, float(smooth_scale), _stream(aux))


def loss_bwd2_blocks(H, W) -> int:
    return _query('clslam_loss_bwd2_blocks', H, W)


def loss_bwd2_pyramid(disps, sel, coef_sel, warped, target, src_m1, src_p1, inv_k, proj, sample_w, ddisp_up, dp_partial,
                      min_depth, max_depth, scales=(0, 4)):
    B, H, W = ddisp_up.shape[1], ddisp_up.shape[-2], ddisp_up.shape[-1]
    _lib.get_lib().call('clslam_loss_bwd2_pyramid_range', _ptr4(disps), _pa(sel, torch.uint8), _p(coef_sel), _p(warped), _p(target),
                        _p(src_m1), _p(src_p1), _p(inv_k), _p(proj), _p(sample_w), _p(ddisp_up), _pa(dp_partial, torch.float64), B, H, W,
                        _nd(min_depth), _nd(max_depth), scales[0], scales[1], _stream(ddisp_up))
