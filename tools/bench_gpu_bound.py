#!/usr/bin/env python
"""How long does the GPU need for one adapt step when the host is infinitely fast?  A spin kernel holds the main stream
(and, through the step's own fork events, the side streams) while the host enqueues the whole step; the time from the end of
the spin to the end of the step is the GPU-side critical path + throughput, with no launch latency in it.  Compared with the
eager step time it says how much a faster launch path could gain at each minibatch size."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'cl-slam_amd'))
import torch, bench
from clslam_hip import synth

H, W = 192, 640
for K in ([int(a) for a in sys.argv[1:]] or (0, 2, 4)):
    B = K + 1
    p = bench.build_predictor(H, W, B)
    p.engine.detached_training = False    # the whole step on the caller's stream: the spin and the two events below live there
    batch = {k: v.cuda() for k, v in synth.make_batch(B, H, W, seed=0).items()}
    for _ in range(60):          # warm clocks: the spin below and a fresh predictor both let them drop
        p.adapt(None, batch, steps=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        p.adapt(None, batch, steps=1)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 40 * 1e3
    spin = int(2.4e9 * (2.5e-3 + 0.5e-3 * B))     # s_sleep cycles: outlasts the host's enqueue of one step (1.5-2 ms)
    ts = []
    for _ in range(8):
        for _ in range(3):
            p.adapt(None, batch, steps=1)
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        torch.cuda._sleep(spin)
        e0.record()
        # adapt() waits for the forward's loss on the host: the spin must outlast the enqueue of the WHOLE step, which it
        # does (the backward and Adam are enqueued before that wait)
        p.adapt(None, batch, steps=1)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f'K={K} (B={B}): eager {eager:.3f} ms/step; GPU-only (host ahead) median {ts[len(ts)//2]:.3f} ms, min {ts[0]:.3f}', flush=True)
    del p
