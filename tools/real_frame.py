#!/usr/bin/env python
"""The REAL adaptation frame of the SLAM loop, timed end to end on one MI355X (review r4 item 2): descriptor pass
(slam.py:143-147) + replay `get` of K samples (slam/replay_buffer.py:186-235,263-291: PNG -> LANCZOS pyramid -> ToTensor ->
colour jitter, per sample and frame) + concatenation with the online sample + adapt() + read-back of the pose and the losses
(slam.py:181-188), with the replay samples built
  (a) the reference's way: on the host with Pillow + torch CPU ops, then uploaded inside adapt(), and
  (b) by clslam_hip.ingest.ReplaySampleBuilder: PNG decode on the host (optionally cached), one uint8 upload, pyramid + jitter
      on the GPU, concatenation on the GPU.
    python tools/real_frame.py [frames=40] [K=4]
KITTI-sized PNGs (376x1241) are written to a temporary directory first.  Measurement tool (no parity claim: the parity of (b)
is tests/test_replay_ingest.py); the host-side jitter in (a) is a plain torch restatement good enough for timing."""
import pickle
import random
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'cl-slam_amd')); sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from PIL import Image  # noqa: E402
import bench  # noqa: E402
from clslam_hip import ingest, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W, RAW_H, RAW_W, STORE = 192, 640, 376, 1241, 12
FRAMES, SCALES = (0, -1, 1), (0, 1, 2, 3)
work = Path(tempfile.mkdtemp())
rng = np.random.default_rng(0)
base = synth.make_batch(1, H, W, seed=0)
files = []
for i in range(STORE):
    sample = {k: v.clone() for k, v in base.items() if k[0] not in ('rgb', 'rgb_aug')}
    for f in FRAMES:
        img = (rng.random((RAW_H // 8, RAW_W // 8, 3)) * 255).astype(np.uint8)
        img = np.asarray(Image.fromarray(img).resize((RAW_W, RAW_H), Image.BICUBIC))
        png = work / f's{i}_{f}.png'
        Image.fromarray(img).save(png)
        sample['rgb', f] = png
    fn = work / f'kitti_{i:05}.pkl'
    with open(fn, 'wb') as fh:
        pickle.dump(sample, fh)
    files.append(fn)


def gray(x):
    return (0.2989 * x[..., 0:1, :, :] + 0.587 * x[..., 1:2, :, :] + 0.114 * x[..., 2:3, :, :])


def host_jitter(x, order, f):        # timing stand-in for torchvision's tensor ops on the CPU
    for op in order:
        if op == 0:
            x = (f[0] * x).clamp(0, 1)
        elif op == 1:
            x = (f[1] * x + (1 - f[1]) * gray(x).mean(dim=(-3, -2, -1), keepdim=True)).clamp(0, 1)
        elif op == 2:
            x = (f[2] * x + (1 - f[2]) * gray(x)).clamp(0, 1)
        else:                        # hue: HSV round trip (same operation count as functional_tensor.adjust_hue)
            mx, mn = x.max(-3).values, x.min(-3).values
            cr = mx - mn
            s = cr / torch.where(cr == 0, torch.ones_like(mx), mx)
            d = torch.where(cr == 0, torch.ones_like(cr), cr)
            r, g, b = x.unbind(-3)
            rc, gc, bc = (mx - r) / d, (mx - g) / d, (mx - b) / d
            h = (mx == r) * (bc - gc) + ((mx == g) & (mx != r)) * (2 + rc - bc) + ((mx != g) & (mx != r)) * (4 + gc - rc)
            h = (torch.fmod(h / 6 + 1, 1) + f[3]) % 1.0
            i = torch.floor(h * 6); ff = h * 6 - i; i = i.to(torch.int64) % 6
            p_, q_, t_ = (mx * (1 - s)).clamp(0, 1), (mx * (1 - ff * s)).clamp(0, 1), (mx * (1 - s * (1 - ff))).clamp(0, 1)
            sel = lambda *c: torch.stack(c, -3).gather(-3, i.unsqueeze(-3)).squeeze(-3)     # noqa: E731
            x = torch.stack((sel(mx, q_, p_, p_, t_, mx), sel(t_, mx, mx, q_, p_, p_), sel(p_, p_, t_, mx, mx, q_)), -3)
    return x


def reference_get(fns):
    """slam/replay_buffer.py:263-291 on the host, K samples concatenated like `get` does"""
    out = None
    for fn in fns:
        order, f = ingest.draw_color_jitter()
        with open(fn, 'rb') as fh:
            data = pickle.load(fh)
        for frame in FRAMES:
            rgb = Image.open(data['rgb', frame]).convert('RGB')
            for s in SCALES:
                rgb = rgb.resize((W >> s, H >> s), Image.LANCZOS)
                t = torch.from_numpy(np.asarray(rgb, dtype=np.float32).transpose(2, 0, 1) / 255.0).unsqueeze(0)
                data['rgb', frame, s] = t
                data['rgb_aug', frame, s] = host_jitter(t, order, f)
            del data['rgb', frame]
        out = data if out is None else {k: torch.cat([out[k], data[k]]) for k in out}
    return out


p = bench.build_predictor(H, W, 1 + K)
p.optimizer.param_groups[0]['lr'] = 1e-6
dev = p.device
online_host = {k: v.pin_memory() for k, v in synth.make_batch(1, H, W, seed=3).items()}
builders = {'gpu ingest, PNGs decoded in one thread': ingest.ReplaySampleBuilder(H, W, SCALES, FRAMES, dev, decode_threads=1),
            'gpu ingest': ingest.ReplaySampleBuilder(H, W, SCALES, FRAMES, dev),            # 16 decode threads (default)
            'gpu ingest, decoded frames cached': ingest.ReplaySampleBuilder(H, W, SCALES, FRAMES, dev, cache_frames=3 * STORE)}
pick = np.random.default_rng(1)


DETAIL = {}


def frame(mode, detail=False):
    sync = torch.cuda.synchronize if detail else (lambda: None)
    tick = time.perf_counter
    ta = tick()
    online = dict(online_host)
    feat = p.models['depth_encoder'](online['rgb', 0, 0].to(dev))[4].mean(-1).mean(-1).cpu()      # slam.py:143-147
    sync()
    fns = [files[i] for i in pick.choice(STORE, K, replace=False)]
    t0 = time.perf_counter()
    if mode == 'reference get (host)':
        replay = reference_get(fns)
        t1 = time.perf_counter()
        training = {k: torch.cat([online[k], replay[k]]) for k in online if k in replay}              # slam.py:300-309
    else:
        datas = builders[mode].get_many(fns)
        replay = {k: torch.cat([d[k] for d in datas]) for k in datas[0]}
        t1 = time.perf_counter()
        training = ingest.cat_dict(online, replay, dev)
    sync()
    t2 = tick()
    out, losses = p.adapt(None, training, steps=1)
    sync()
    t3 = tick()
    T = out['cam_T_cam', 0, 1][0, :].squeeze().cpu().numpy()
    vals = {k: float(v) for k, v in losses.items()}
    assert vals['loss'] == vals['loss'] and abs(T).max() < 1e3 and feat.shape == (1, 512)
    if detail:
        for name, v in (('descriptor', t0 - ta), ('get', t1 - t0), ('cat', t2 - t1), ('adapt', t3 - t2), ('read-back', tick() - t3)):
            DETAIL[name] = DETAIL.get(name, 0.0) + v
    return t1 - t0


print(f'real frame @{H}x{W}, K = {K} replay samples from a store of {STORE} ({RAW_H}x{RAW_W} PNGs), {N} frames per mode')
for mode in ('reference get (host)', 'gpu ingest, PNGs decoded in one thread', 'gpu ingest', 'gpu ingest, decoded frames cached'):
    random.seed(0)
    for _ in range(3 if mode == 'reference get (host)' else STORE):        # warm-up (and fill the decoded-frame cache)
        frame(mode)
    torch.cuda.synchronize()
    n = max(4, N // 4) if mode == 'reference get (host)' else N
    t0 = time.perf_counter()
    tg = 0.0
    for _ in range(n):
        tg += frame(mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f'  {mode:36s}: {dt:8.2f} ms per frame  ({1e3 / dt:6.1f} frames/s)   of which `get`: {tg / n * 1e3:7.2f} ms')
    if mode != 'reference get (host)':
        DETAIL.clear()
        for _ in range(10):
            frame(mode, detail=True)
        print('      stage by stage, a synchronize between the stages (ms):', {k: round(v / 10 * 1e3, 2) for k, v in DETAIL.items()})
