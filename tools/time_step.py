#!/usr/bin/env python
"""Break the adapt step down: eager forward / backward / adam vs hipGraph replay, host vs device time."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'cl-slam_amd')); sys.path.insert(0, str(ROOT))
import torch
import bench
from clslam_hip import synth

H, W, B = 192, 640, 5
p = bench.build_predictor(H, W, B)
dev = p.device
batch = {k: v.to(dev) for k, v in synth.make_batch(B, H, W, seed=0).items()}
eng = p.engine
p._set_adapt(); eng.pack_if_needed()
sw, smw = p._sample_weights(B, None)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    return t_host * 1e3, t_all * 1e3


print('forward  (host ms, total ms):', timeit(lambda: eng.forward(batch, train=True, sample_w=sw, smooth_w=smw)))
print('backward (host ms, total ms):', timeit(lambda: eng.backward(B)))
print('adam     (host ms, total ms):', timeit(lambda: eng.adam(1e-4)))
for ss in ('1', '0'):
    eng.use_side_stream = ss == '1'
    print(f'fwd+bwd side_stream={ss}:', timeit(lambda: (eng.forward(batch, train=True, sample_w=sw, smooth_w=smw), eng.backward(B))))
eng.use_side_stream = True
out = eng.train_step_graphed(batch, sample_w=sw, smooth_w=smw, noise=None)
st = list(eng._graphs.values())[0]
print('graph replay only:', timeit(lambda: st.graph.replay()))
print('graphed step incl. copies/clones:', timeit(lambda: eng.train_step_graphed(batch, sample_w=sw, smooth_w=smw, noise=None)))
eng.use_side_stream = False
eng._graphs.clear()
eng.train_step_graphed(batch, sample_w=sw, smooth_w=smw, noise=None)
st = list(eng._graphs.values())[0]
print('graph replay only (single stream capture):', timeit(lambda: st.graph.replay()))

# ---- whole adapt() call, both modes -----------------------------------------------------------------
eng.use_side_stream = True
eng._graphs.clear()
for mode in ('0', '1', '0', '1'):
    os.environ['CLSLAM_HIPGRAPH'] = mode
    print(f'adapt() hipgraph={mode}:', timeit(lambda: p.adapt(None, batch, steps=1), n=30))
t0 = time.perf_counter()
for _ in range(50):
    p._set_adapt()
print('_set_adapt host ms:', (time.perf_counter() - t0) / 50 * 1e3)
t0 = time.perf_counter()
for _ in range(50):
    eng.pack_if_needed()
print('pack_if_needed host ms:', (time.perf_counter() - t0) / 50 * 1e3)
t0 = time.perf_counter()
for _ in range(50):
    p._sample_weights(B, None)
torch.cuda.synchronize()
print('_sample_weights ms:', (time.perf_counter() - t0) / 50 * 1e3)
