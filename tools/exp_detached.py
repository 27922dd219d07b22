"""Experiment: host-side cost of the detached training step at small batches (eager adapt() loop, no read-back)."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'cl-slam_amd'))
import torch, bench
from clslam_hip import synth
H, W = 192, 640
for K in (0, 4):
    B = K + 1
    p = bench.build_predictor(H, W, B)
    batch = {k: v.cuda() for k, v in synth.make_batch(B, H, W, seed=0).items()}
    for mode in ('attached', 'detached', 'attached', 'detached'):
        p.engine.detached_training = mode == 'detached'
        for _ in range(30):
            p.adapt(None, batch, steps=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            p.adapt(None, batch, steps=1)
        torch.cuda.synchronize()
        print(f'K={K} {mode}: {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/step  (NO_RECORD={os.environ.get("CLSLAM_EXP_NO_RECORD")})', flush=True)
    if K == 0:
        import cProfile, pstats
        p.engine.detached_training = True
        pr = cProfile.Profile(); pr.enable()
        for _ in range(40):
            p.adapt(None, batch, steps=1)
        pr.disable(); torch.cuda.synchronize()
        st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(14)
