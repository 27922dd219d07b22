#!/usr/bin/env python
"""Time clslam_reduce_multi alone on the real per-layer partial table of a 192x640 B=5 adapt step."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'cl-slam_amd')); sys.path.insert(0, str(ROOT))
import torch
import bench
from clslam_hip import ops, synth

H, W, B = 192, 640, 5
p = bench.build_predictor(H, W, B)
batch = {k: v.to(p.device) for k, v in synth.make_batch(B, H, W, seed=0).items()}
for _ in range(2):
    p.adapt(None, batch, steps=1)
torch.cuda.synchronize()
t = p.engine._ws[B].train
tot = sum(n * s for _, _, n, s in t.items) * 4 / 1e6
print(f'{len(t.items)} items, {tot:.1f} MB of partials')
for it in sorted(t.items, key=lambda x: -x[2] * x[3])[:8]:
    print('  n', it[2], 'splits', it[3], f'{it[2] * it[3] * 4 / 1e6:.1f} MB')
for bpi in (96, 192, 384, 768):
    for _ in range(3):
        ops.reduce_multi(t.table, len(t.items), p.engine.g, blocks_per_item=bpi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20):
        ops.reduce_multi(t.table, len(t.items), p.engine.g, blocks_per_item=bpi)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f'blocks_per_item {bpi}: {us:.1f} us  -> {tot / us * 1e-3 * 1e3:.0f} GB/s')
# one item at a time
for it in sorted(t.items, key=lambda x: -x[2] * x[3]):
    tab = ops.make_reduce_table([it], p.device)
    for _ in range(3):
        ops.reduce_multi(tab, 1, p.engine.g, blocks_per_item=384)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20):
        ops.reduce_multi(tab, 1, p.engine.g, blocks_per_item=384)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f'  n {it[2]} splits {it[3]}: {us:.1f} us -> {it[2] * it[3] * 4 / us * 1e-3:.0f} GB/s')
