#!/usr/bin/env python
"""Phase timeline of conv_wino.hip (build with CLSLAM_HIPCC_EXTRA=-DCLSLAM_WINO_TRACE=1): per-workgroup s_memtime stamps."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'cl-slam_amd'))
from clslam_hip import ops  # noqa: E402
dev = torch.device('cuda:0')
B, H, W, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
CFG = 40
x = torch.randn(B, H, W, C, device=dev)
w = torch.randn(C, 9, C, device=dev) * 0.05
out = torch.empty(B, H, W, C, device=dev)
ws = torch.zeros(48 << 20, dtype=torch.uint8, device=dev)
u = ops.wino_weight_transform(w)
for _ in range(3):
    ops.conv2d(x, w, out, ksize=3, act=1, config=CFG, workspace=ws, weight_wino=u)
torch.cuda.synchronize()
G = 256
off = (64 << 10) + G * 64 * 4 * 64 * 4
tr = ws[off:off + G * 64 * 8].view(torch.int64).view(G, 64).cpu()
t0 = tr[:, 0].min()
tr = (tr - t0).clamp_min(-1)
import numpy as np
a = tr.numpy().astype(np.float64) / 100.0      # s_memtime ticks = shader cycles (tools/micro/mfma_clock.hip): columns in units of 100 cycles
print('columns: start | prologue landed | prologue done | per unit: loop done, finish done, barrier passed  (us)')
for g in (0, 1, 2, 3, 100, 101, 255):
    row = a[g]
    n = int((tr[g] >= 0).sum())
    print(f'wg {g:3d}:', ' '.join(f'{v - row[0]:6.1f}' for v in row[:min(n, 30)]))
end = np.array([a[g][max(0, int((tr[g] > 0).sum()) - 1)] for g in range(G)])
print('start  min/median/max:', a[:, 0].min(), np.median(a[:, 0]), a[:, 0].max())
print('prologue landed median:', np.median(a[:, 1] - a[:, 0]), ' prologue transform median:', np.median(a[:, 2] - a[:, 1]))
print('end    min/median/max:', end.min(), np.median(end), end.max())
d = a[:, 3:24]
loop = d[:, 0::3] - np.concatenate([a[:, 2:3], d[:, 2::3][:, :-1]], 1)
fin = d[:, 1::3] - d[:, 0::3]
bar = d[:, 2::3] - d[:, 1::3]
valid = d[:, 2::3] > 0
print('MFMA loop per unit, median by unit index:', [round(float(np.median(loop[:, i][valid[:, i]])), 2) for i in range(loop.shape[1]) if valid[:, i].any()])
print('finish     per unit, median / max        :', [(round(float(np.median(fin[:, i][valid[:, i]])), 2), round(float(fin[:, i][valid[:, i]].max()), 2)) for i in range(fin.shape[1]) if valid[:, i].any()])
print('wait+barrier per unit, median / max      :', [(round(float(np.median(bar[:, i][valid[:, i]])), 2), round(float(bar[:, i][valid[:, i]].max()), 2)) for i in range(bar.shape[1]) if valid[:, i].any()])
