"""Diagnostic (GPU): is the read-out of the bilinear cells (clslam_warp_cells_pyramid) the decision the forward took?  Every
synthesised pixel must lie in the hull of the four taps of its read-out cell.  (The sampling position is computed by ONE
contraction-free function chain shared by the forward, the loss backward and the read-out -- geometry_dev.h -- so the three
agree to the bit by construction; before round 3's fix the backward could floor a sample within one ulp of a cell boundary
to the neighbouring cell.)"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
    sys.path.insert(0, str(p))
import torch
from clslam_hip import ops, synth
from predictor_util import make_predictor

H, W, B = 192, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 5
p = make_predictor(H, W, B)
batch = synth.make_batch(B, H, W, seed=5)
noise = synth.make_noise(B, H, W, seed=15)
p.set_tie_break_noise(noise)
out, losses = p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
ws = p.engine._ws[B]
c = torch.empty(4, 2, B, H, W, dtype=torch.int32, device='cuda')
ops.warp_cells_pyramid(ws.disp, ws.ctx.Kinv, ws.P, c, p.min_depth, p.max_depth)
cells = [c.cpu()]
# forward consistency: nearest-corner reconstruction of warped from the cells is not possible without the weights; instead
# check that the warped value lies within the convex hull of the four taps of the readout cell (violations = readout != forward)
c = cells[0].long()
x0, y0 = c & 0xFFF, (c >> 12) & 0xFFF
viol = 0
for fi, f in enumerate((-1, 1)):
    src = batch['rgb', f, 0]
    flat = src.reshape(B, 3, H * W)
    for s in range(4):
        xa, ya = x0[s, fi], y0[s, fi]
        xb, yb = (xa + 1).clamp(max=W - 1), (ya + 1).clamp(max=H - 1)
        taps = torch.stack([torch.gather(flat, 2, (yy * W + xx).reshape(B, 1, -1).expand(B, 3, -1)) for yy, xx in ((ya, xa), (ya, xb), (yb, xa), (yb, xb))])
        lo, hi = taps.min(0).values, taps.max(0).values
        wv = out['rgb', f, s].cpu().reshape(B, 3, -1)
        viol += int(((wv < lo - 1e-6) | (wv > hi + 1e-6)).sum())
print('warped values outside the hull of their readout cell:', viol)
