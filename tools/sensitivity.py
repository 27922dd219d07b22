#!/usr/bin/env python
"""What is each kernel class worth INSIDE the step?  The step overlaps three streams and is bound by the sum of its kernels
rather than by any chain, so a kernel's stand-alone time says little about what speeding it up would buy.  This tool removes
one class of launches at a time (results are garbage, the NaN guard is disarmed: TIMING ONLY) and reports how much the
step shrinks: the marginal cost of the class, an upper bound for any optimisation of it."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'cl-slam_amd'))
import torch, bench
from clslam_hip import ops, synth
import depth_pose_prediction.depth_pose_prediction as dpp

H, W, B = 192, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 5
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # optimizer steps per adapt() call (5: the reference's shipped setting)
dpp.DepthPosePrediction._raise_on_nan = lambda self, *a, **k: None
p = bench.build_predictor(H, W, B)
batch = {k: v.cuda() for k, v in synth.make_batch(B, H, W, seed=0).items()}
orig = {n: getattr(ops, n) for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith('_')}


def timed(n=40, warm=12):
    for _ in range(warm):
        p.adapt(None, batch, steps=S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        p.adapt(None, batch, steps=S)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def skip(names=(), conv_pred=None):
    for n, f in orig.items():
        setattr(ops, n, f)
    for n in names:
        setattr(ops, n, lambda *a, **k: None)
    if conv_pred is not None:
        def conv2d(src_a, weight, out, **kw):
            if conv_pred(src_a, weight, out, kw):
                return out
            return orig['conv2d'](src_a, weight, out, **kw)
        ops.conv2d = conv2d


cin = lambda a, w: w.shape[2]
classes = [
    ('nothing (base)', (), None),
    ('weight gradients (conv_wgrad*, dispconv_wgrad, colsum)', ('conv_wgrad', 'conv_wgrad_patch', 'dispconv_wgrad', 'colsum'), None),
    ('fold_act_grad (10 launches)', ('fold_act_grad',), None),
    ('loss stage forward (warp_fwd, photo_automask)', ('warp_fwd_pyramid', 'photo_automask_pyramid', 'photo_automask_pyramid_rng'), None),
    ('loss stage backward (loss_bwd2, disp_grad)', ('loss_bwd2_pyramid', 'disp_grad_pyramid'), None),
    ('stems + maxpools', ('stem_conv', 'maxpool3x3s2'), None),
    ('identity maps + weight transposes', ('photo_map', 'weight_transpose'), None),
    ('reduce_multi', ('reduce_multi',), None),
    ('convs with Cin >= 256 (the stream-K class)', (), lambda a, w, o, k: cin(a, w) >= 256),
    ('convs with 64 <= Cin < 256', (), lambda a, w, o, k: 64 <= cin(a, w) < 256),
    ('convs with Cin < 64 (high resolution)', (), lambda a, w, o, k: cin(a, w) < 64),
    ('forward convs of the frozen encoders (scale/shift folded BN)', (), lambda a, w, o, k: k.get('scale') is not None),
    ('dgrad convs (pad = 2)', (), lambda a, w, o, k: k.get('pad') == 2),
]
base = None
for label, names, pred in classes:
    skip(names, pred)
    ms = min(timed(), timed())
    if base is None:
        base = ms
    print(f'{label:62s} {ms:7.3f} ms   {ms - base:+7.3f}', flush=True)
