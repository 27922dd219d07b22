#!/usr/bin/env python
"""Diagnostic (GPU box): which trainable tensors hold the step-1 updates that land on the other side of the float64 oracle's
(`w_flipped` of tests/test_teacher_forced_steps.py), for the HIP path and for the oracle's own fp32 run; and how large the
float64 gradient is there.  python tools/diag_flips.py [H W B]"""
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
    sys.path.insert(0, str(p))
import torch  # noqa: E402
from clslam_hip import synth  # noqa: E402
from clslam_hip.engine import TrainableLayout  # noqa: E402
from emu_util import use_backend  # noqa: E402
from helpers import make_oracle, oracle_grads  # noqa: E402
from predictor_util import make_predictor  # noqa: E402

H, W, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (192, 640, 1)
LR = 1e-4
use_backend('hip')
batch = synth.make_batch(B, H, W, seed=33)
noise = synth.make_noise(B, H, W, seed=53)
b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
n64 = {s: v.double() for s, v in noise.items()}
o64 = make_oracle(H, W, B).to_double()
_, _, g64 = oracle_grads(o64, b64, n64)
o32 = make_oracle(H, W, B)
_, _, g32 = oracle_grads(o32, batch, noise)
p = make_predictor(H, W, B)
p.set_tie_break_noise(noise)
p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
eng = p.engine
ghip = {name: TrainableLayout.to_reference(eng.g[off:off + math.prod(shape)], shape).cpu().double() for name, off, shape in eng.layout.entries}
print(f'{"tensor":46s} {"n":>8s} {"sign flips hip":>14s} {"torch32":>8s} {"|g64| median":>12s} {"|g64| at hip flips (median)":>28s} {"|g_hip - g64| there":>20s} {"|g32 - g64| there":>18s}')
tot = [0, 0, 0]
for name, g in g64.items():
    gh, g3 = ghip[name], g32[name].double()
    fh = (torch.sign(gh) != torch.sign(g)) & (g != 0)
    f3 = (torch.sign(g3) != torch.sign(g)) & (g != 0)
    tot[0] += int(fh.sum()); tot[1] += int(f3.sum()); tot[2] += g.numel()
    if int(fh.sum()) + int(f3.sum()) == 0:
        continue
    at = g.abs()[fh]
    print(f'{name:46s} {g.numel():8d} {int(fh.sum()):14d} {int(f3.sum()):8d} {float(g.abs().median()):12.2e} '
          f'{float(at.median()) if at.numel() else 0:28.2e} {float((gh - g).abs()[fh].median()) if at.numel() else 0:20.2e} '
          f'{float((g3 - g).abs()[fh].median()) if at.numel() else 0:18.2e}')
print('total sign flips: hip', tot[0], 'torch fp32', tot[1], 'of', tot[2], f'-> {tot[0] / tot[2]:.2e} / {tot[1] / tot[2]:.2e}')
