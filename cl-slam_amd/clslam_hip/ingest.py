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


# ======================================================================================================================
# The replay buffer's samples on the GPU (the path that runs on EVERY adapted frame: slam/slam.py:98 builds
# the buffer with do_augmentation=True, slam/replay_buffer.py:186-235 `get` -> :263-291 `_get`)
def jitter_params(draws, device) -> torch.Tensor:
    """draws: one (order, factors) per image -- op ids in application order and the four factors by op id, as
    get_random_color_jitter (datasets/utils.py:236-259) draws them -> the device records clslam_color_jitter_f32 reads."""
    import numpy as np
    rec = np.zeros(len(draws), dtype=[('order', np.int32, 4), ('f', np.float32, 4), ('omf', np.float32, 4)])
    for i, (order, factors) in enumerate(draws):
        o = [int(v) for v in order][:4]
        rec['order'][i] = o + [-1] * (4 - len(o))
        rec['f'][i] = [np.float32(float(v)) for v in factors]
        rec['omf'][i] = [np.float32(1.0 - float(v)) for v in factors]        # 1.0 - ratio in double, then a C float (torch)
    return torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(device)


def color_jitter_tensor(images: torch.Tensor, params: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """images (N,3,h,w) float32 in [0,1] on the library's device, params = jitter_params(...) with N records: torchvision's
    tensor-path adjust_brightness / _contrast / _saturation / _hue chain per image (each with its own contrast mean)."""
    if images.dtype != torch.float32 or images.dim() != 4 or images.shape[1] != 3:
        raise _lib.ClslamError(f'color_jitter_tensor expects (N,3,h,w) float32 images, got {tuple(images.shape)} {images.dtype}')
    images = images.contiguous()
    N, _, h, w = images.shape
    if params.numel() != N * 48:
        raise _lib.ClslamError(f'color_jitter_tensor: {params.numel()} parameter bytes for {N} images (48 each)')
    lib = _lib.get_lib()
    out = torch.empty_like(images) if out is None else out
    partial = torch.empty(max(N, 1) * lib.cdll.clslam_color_jitter_f32_blocks(h, w), device=images.device)
    lib.call('clslam_color_jitter_f32', _p(images), _p(out), _pa(params, torch.uint8), _p(partial), N, h, w, _stream(images))
    return out


def draw_color_jitter(brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=(-.1, .1), rng=None):
    """The draws of datasets/utils.py:236-259 from Python's `random` IN THE REFERENCE'S ORDER (four uniforms, then the shuffle of
    the four-element transform list), so that a run seeded like the reference consumes the same random stream."""
    import random
    rng = random if rng is None else rng
    factors = [rng.uniform(*brightness), rng.uniform(*contrast), rng.uniform(*saturation), rng.uniform(*hue)]
    order = [BRIGHTNESS, CONTRAST, SATURATION, HUE]
    rng.shuffle(order)
    return order, factors


class ReplaySampleBuilder:
    """What ``ReplayBuffer._get`` (slam/replay_buffer.py:263-291) builds per replayed sample -- PNG -> RGB, the LANCZOS pyramid level
    from level, ToTensor, one drawn colour jitter on every level -- with the pixels on the GPU: the host decodes the PNGs
    (Pillow, on a small thread pool: its decoder releases the GIL), the raw uint8 frames cross PCIe ONCE, pyramid and jitter are
    three kernels per level for all samples and frames of a minibatch.

        build = ReplaySampleBuilder(height, width, scales, frames, device)
        build.install(slam.replay_buffer, slam)       # the reference's Slam / ReplayBuffer objects, everything else unchanged
        samples = build.get_many(filenames)           # or by hand: the K samples of a frame in one go

    Non-image entries of the pickled sample (camera matrices, relative distances, index ...) are passed through unchanged.
    """

    def __init__(self, height: int, width: int, scales: Sequence[int] = (0, 1, 2, 3), frames: Sequence[int] = (0, -1, 1),
                 device=None, do_augmentation: bool = True, cache_frames: int = 0, cache_bytes: int = 256 << 20,
                 decode_threads: int = 16) -> None:
        self.height, self.width, self.scales, self.frames = int(height), int(width), tuple(scales), tuple(frames)
        self.device = torch.device('cuda' if device is None else device)
        self.do_augmentation = bool(do_augmentation)
        self.pyramid = ImagePyramid(height, width, tuple(range(max(self.scales) + 1)))
        # opt-in: keep the DECODED uint8 frames of the last `cache_frames` image files on the host (a replay buffer re-serves
        # the same few hundred samples: a 1241x376 PNG costs ~7 ms to decode, its 1.4 MB cost nothing to keep).  Bounded by
        # count AND by bytes (`cache_bytes`, page-locked memory on a GPU build): the oldest entries go first.
        self.cache_frames = int(cache_frames)
        self.cache_bytes = int(cache_bytes)
        self._cached_bytes = 0
        self._decoded: Dict = {}
        # The K x 3 PNGs of a minibatch are decoded concurrently: Image.open(...).convert('RGB') (replay_buffer.py:271) spends its
        # time in Pillow's C decoder with the GIL released.  0 / 1: in the calling thread, like the reference.
        self.decode_threads = int(decode_threads)
        self._pool = None

    # ------------------------------------------------------------------------------------------------------------------
    def install(self, replay_buffer, slam=None) -> 'ReplaySampleBuilder':
        """Switch the REFERENCE's objects to this builder -- instance attributes only, their classes stay untouched:

        * ``replay_buffer.get`` still runs the reference's own sampling code (slam/replay_buffer.py:186-227: the choice of the
          files is its business), but with ``_get`` recording the file names instead of decoding them one by one; the K samples
          are then built by ONE get_many() -- one upload, one pyramid / jitter pass -- and concatenated per key exactly like
          replay_buffer.py:229-233.  The colour-jitter draws come from Python's ``random`` in the reference's order (one draw
          per file, the sampler draws from its own numpy generator before any of them), so a seeded run consumes the same
          random streams.
        * ``replay_buffer._get`` (direct callers) becomes get_one.
        * ``slam._cat_dict`` (slam/slam.py:300-309 joins the online sample, host tensors, with the replay minibatch) becomes
          cat_dict below: torch.cat refuses mixed devices, the online entries are uploaded first.
        """
        builder = self
        reference_get = replay_buffer.get
        replay_buffer._get = self.get_one

        def get(*args, **kwargs):
            names = []

            def record(filename, include_batch=True):
                names.append((filename, include_batch))
                return {}                                 # nothing to concatenate: replay_buffer.py:229-233 loops over its keys
            replay_buffer._get = record
            try:
                stub = reference_get(*args, **kwargs)
            finally:
                replay_buffer._get = builder.get_one
            if not names:
                return stub
            datas = builder.get_many([n for n, _ in names], include_batch=names[0][1])
            out = datas[0]
            for data in datas[1:]:
                for key in out:
                    out[key] = torch.cat([out[key], data[key]])
            return out
        replay_buffer.get = get
        if slam is not None:
            slam._cat_dict = cat_dict
        return self

    def _decode_file(self, path):
        import numpy as np
        from PIL import Image
        return torch.from_numpy(np.asarray(Image.open(path).convert('RGB')).copy())       # replay_buffer.py:271

    def _keep(self, key, img):
        if self.cache_frames <= 0:
            return img
        if self.device.type == 'cuda':
            img = img.pin_memory()                                  # page-locked ONCE (~40 ms): later uploads are asynchronous DMAs
        nbytes = img.numel()
        while self._decoded and (len(self._decoded) >= self.cache_frames or self._cached_bytes + nbytes > self.cache_bytes):
            old = self._decoded.pop(next(iter(self._decoded)))
            self._cached_bytes -= old.numel()
        if nbytes <= self.cache_bytes:
            self._decoded[key] = img
            self._cached_bytes += nbytes
        return img

    def _decode_all(self, paths):
        """decoded uint8 frames of `paths` (cache hits first, the misses concurrently)"""
        keys = [str(p) for p in paths]
        out = {k: self._decoded[k] for k in keys if k in self._decoded}
        misses = [k for k in dict.fromkeys(keys) if k not in out]
        if len(misses) > 1 and self.decode_threads > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self.decode_threads, thread_name_prefix='clslam-png')
            imgs = list(self._pool.map(self._decode_file, misses))
        else:
            imgs = [self._decode_file(k) for k in misses]
        for k, img in zip(misses, imgs):
            out[k] = self._keep(k, img)
        return [out[k] for k in keys]

    def get_many(self, filenames, include_batch: bool = True, rng=None):
        """-> one dict per file, exactly the entries `_get(filename, include_batch)` returns."""
        import pickle

        import numpy as np
        datas, paths, draws = [], [], []
        for fn in filenames:
            # the jitter is drawn BEFORE the file is read (replay_buffer.py:264-268): same random stream as the reference
            draw = draw_color_jitter(rng=rng) if self.do_augmentation else None
            with open(fn, 'rb') as f:
                data = pickle.load(f)
            datas.append(data)
            for frame in self.frames:
                paths.append(data['rgb', frame])
                draws.append(draw)
        if not paths:
            return []
        raws = self._decode_all(paths)
        if len({tuple(r.shape) for r in raws}) != 1:
            raise _lib.ClslamError('replay samples with different raw image sizes in one batch')
        # one device staging block, one asynchronous copy per (page-locked) frame
        stage = torch.empty((len(raws),) + tuple(raws[0].shape), dtype=torch.uint8, device=self.device)
        for i, r in enumerate(raws):
            stage[i].copy_(r, non_blocking=True)
        levels = self.pyramid(stage)                                 # {s: (n_img,3,h,w) float}, bit-exact vs Pillow
        aug = levels
        if self.do_augmentation:
            params = jitter_params(draws, self.device)
            aug = {s: color_jitter_tensor(levels[s], params) for s in self.scales}
        nf = len(self.frames)
        for i, data in enumerate(datas):
            for j, frame in enumerate(self.frames):
                k = i * nf + j
                for s in self.scales:
                    rgb, rga = levels[s][k], aug[s][k]
                    data['rgb', frame, s] = rgb.unsqueeze(0) if include_batch else rgb
                    data['rgb_aug', frame, s] = rga.unsqueeze(0) if include_batch else rga
                del data['rgb', frame]
            if not include_batch:                                # replay_buffer.py:286-289
                for key in data:
                    if not ('rgb' in key or 'rgb_aug' in key):
                        data[key] = data[key].squeeze(0)
        return datas

    def get_one(self, filename, include_batch: bool = True):
        """``ReplayBuffer._get(filename, include_batch=True)``"""
        return self.get_many([filename], include_batch)[0]


def cat_dict(online: Dict, replay: Dict, device=None) -> Dict:
    """slam/slam.py:300-309 (`_cat_dict`) for a replay dict whose image tensors already live on the GPU: entries present in both
    dicts are concatenated there (the online sample's entries are uploaded; torch.cat refuses mixed devices)."""
    out = {}
    for k in online:
        if k in replay:
            a, b = online[k], replay[k]
            dev = device if device is not None else (b.device if b.device.type != 'cpu' else a.device)
            out[k] = torch.cat([a.to(dev, non_blocking=True), b.to(dev, non_blocking=True)])
    return out
