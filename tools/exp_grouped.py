"""Experiment (timing only): what would ONE grouped launch per layer for the two frozen encoders buy?  The depth encoder (B
images) and the pose encoder (2B image pairs) have equal layer shapes and different weights; here the upper bound is measured
without writing the grouped kernel: (A) the engine's schedule -- depth encoder on the main stream, pose encoder on the side
stream, concurrently; (B) ONE encoder pass over 3B images with one weight set (same flops, same launch shapes as grouped
launches would have); (C) the two encoders one after the other on one stream.   python tools/exp_grouped.py [B=5]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'cl-slam_amd'))
import torch, bench
from clslam_hip import ops, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H, W = 192, 640
p = bench.build_predictor(H, W, B)
eng = p.engine
batch = {k: v.cuda() for k, v in synth.make_batch(3 * B, H, W, seed=0).items()}
p.adapt(None, {k: v[:B].contiguous() for k, v in batch.items()})
torch.cuda.synchronize()
a0, am, ap = (batch['rgb_aug', f, 0] for f in (0, -1, 1))
ws = eng.workspace(B)
big = eng._enc_bufs(3 * B)
main, side = torch.cuda.current_stream(), eng.side_stream
eng._conv_workspace(main.cuda_stream)


def sched_a():
    side.wait_stream(main)
    eng._encoder(eng.enc['depth_encoder'], ws.denc, B, [(a0[:B], None, 0, B)])
    with ops.launch_on(side):
        eng._encoder(eng.enc['pose_encoder'], ws.penc, 2 * B, [(am[:B], a0[:B], 0, B), (a0[:B], ap[:B], B, B)], stream=side)
    main.wait_stream(side)


def sched_b():
    eng._encoder(eng.enc['pose_encoder'], big, 3 * B, [(am, a0, 0, 3 * B)])


def sched_c():
    eng._encoder(eng.enc['depth_encoder'], ws.denc, B, [(a0[:B], None, 0, B)])
    eng._encoder(eng.enc['pose_encoder'], ws.penc, 2 * B, [(am[:B], a0[:B], 0, B), (a0[:B], ap[:B], B, B)])


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, fn in (('A two streams (engine)', sched_a), ('B one pass over 3B images', sched_b), ('C two encoders, one stream', sched_c),
                 ('A again', sched_a), ('B again', sched_b)):
    print(f'B={B}: {name:32s} {timeit(fn):.3f} ms', flush=True)
