#!/usr/bin/env python
"""SURVEY.md 8(f) rank 2 -- what memoising the replay-buffer descriptor pass (slam/slam.py:143-147:
`models['depth_encoder'](online_image)[4]`, one B=1 encoder forward per frame BEFORE adapt()) could save, measured:

  a) the descriptor pass as slam.py runs it (upload of one image + encoder + pooled feature to the host);
  b) adapt(steps=1) on B samples vs on B-1 samples: the most a skipped sample-0 depth-encoder pass could give back
     (an upper bound: it also drops sample 0's decoder, loss and backward);
  c) the depth encoder alone at batch B vs B-1 (what a memo hit would actually skip);
  d) what a content check costs: the online image reaches the descriptor pass as a DEVICE tensor and adapt() as a row of a
     freshly concatenated HOST tensor (slam.py:300-309), so identity cannot match; comparing contents needs either a device
     compare + host sync or a host memcmp of the 1.5 MB plane."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'cl-slam_amd')); sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from clslam_hip import synth  # noqa: E402

H, W, B = 192, 640, 5


def t(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


p = bench.build_predictor(H, W, B)
dev = p.device
full = synth.make_batch(B, H, W, seed=0)
img_host = full['rgb', 0, 0][:1].pin_memory()


def descriptor():
    p._set_eval()
    with torch.no_grad():
        x = img_host.to(dev, non_blocking=True)
        f = p.models['depth_encoder'](x)[4].detach()
        return f.mean(-1).mean(-1).cpu().numpy()


print(f'a) descriptor pass as slam.py:143-147 runs it: {t(descriptor):.3f} ms')
for n in (B, B - 1):
    batch = {k: v[:n].to(dev) for k, v in full.items()}
    pn = p if n == B else bench.build_predictor(H, W, n)
    print(f'b) adapt(steps=1) on {n} samples: {t(lambda: pn.adapt(None, batch, steps=1)):.3f} ms')
for n in (B, B - 1):
    x = full['rgb_aug', 0, 0][:n].to(dev)
    print(f"c) depth encoder forward alone, batch {n}: {t(lambda: p.engine.run_encoder('depth_encoder', x)):.3f} ms")
a = full['rgb', 0, 0][:1].to(dev)
b = a.clone()
print(f'd) device compare + host sync of one 3x{H}x{W} plane: {t(lambda: bool(torch.equal(a, b))):.3f} ms')
ah, bh = full['rgb', 0, 0][:1].clone(), full['rgb', 0, 0][:1].clone()
print(f'd) host compare of the same plane: {t(lambda: bool(torch.equal(ah, bh)), n=50):.3f} ms')

# ---- round 3: the frame as slam.py runs it when the replay buffer is off (K = 0) or adaptation is off (slam.py:178) --------
# descriptor pass (slam.py:143-147) + adapt() on the single online triplet, with and without the descriptor memo
# (Engine._memo_lookup: the adapt forward takes the descriptor pass's features when its network input has the same content).
print('--- single-triplet frames, descriptor pass + adapt, memo on / off (ms per frame)')
one = synth.make_batch(1, H, W, seed=0)
for k in list(one):
    if k[0] == 'rgb_aug':
        one[k] = one['rgb', k[1], k[2]].clone()          # the online dataset does not augment (datasets/utils.py:25)
host = {k: v.pin_memory() for k, v in one.items()}
p1 = bench.build_predictor(H, W, 1)


def frame(train):
    online = dict(host)
    p1._set_eval()
    with torch.no_grad():
        x = online['rgb', 0, 0].to(dev)
        f = p1.models['depth_encoder'](x)[4].detach().mean(-1).mean(-1).cpu().numpy()
    out, losses = p1.adapt(online, online if train else None)
    T = out['cam_T_cam', 0, 1][0, :].squeeze().cpu().detach().numpy()
    return f, T, {k: float(v.squeeze().cpu().detach().numpy()) for k, v in losses.items()}


for train in (True, False):
    res = {}
    for memo in (False, True, False, True):
        p1.engine.descriptor_memo = memo
        h0 = p1.engine.memo_hits
        ms = t(lambda: frame(train), n=40, warm=8)
        res.setdefault(memo, []).append(ms)
        hits = p1.engine.memo_hits - h0
    off, on = min(res[False]), min(res[True])
    print(f"{'K=0 adapt(online, online)' if train else 'no adaptation: adapt(online, None)'}: memo off {off:.3f}  on {on:.3f}  "
          f"({(off - on) / off * 100:.1f} % of the frame; hits in the last run {hits}/48)")
