#!/usr/bin/env python
"""Per-WAVE phase timeline of conv3x3_wino8_kernel (variant built with -DCLSLAM_WINO_TRACE=3, tools/build_variant.py):

    CLSLAM_TOOL_LIB=wtrace python tools/wino_trace_waves.py B H W C [groups]

every wave's lane 0 stamps s_memtime (shader cycles) at: start | prologue done | per unit: input transform done, MFMA loop done,
first-half MFMAs done, middle barrier passed, second-half MFMAs done, finish() done, end barrier passed."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import _variant  # noqa: F401,E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from clslam_hip import ops  # noqa: E402

dev = torch.device('cuda:0')
B, H, W, C = (int(a) for a in sys.argv[1:5])
G = int(sys.argv[5]) if len(sys.argv) > 5 else 256
os.environ['CLSLAM_WINO_GROUPS'] = str(G)
x = torch.randn(B, H, W, C, device=dev)
w = torch.randn(C, 9, C, device=dev) * 0.05
out = torch.empty(B, H, W, C, device=dev)
ws = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
u = ops.wino_weight_transform(w)
for _ in range(3):
    ops.conv2d(x, w, out, ksize=3, act=1, config=40, workspace=ws, weight_wino=u)
torch.cuda.synchronize()
off = (64 << 10) + G * 64 * 4 * 64 * 4
tr = ws[off:off + G * 8 * 64 * 8].view(torch.int64).view(G, 8, 64).cpu().numpy().astype(np.float64)
t0 = tr[:, :, 0].min()
a = tr - t0
NS = 5        # stamps per unit: patch DMA issue + input transform | 16 positions (+ U DMA issue) | finish() | DMA wait | barrier
names = ('T', 'M', 'F', 'W', 'B')
print(f'# B={B} {H}x{W} C={C} G={G}; cycles per phase of a unit: T patch DMA issue + input transform, M the 16 positions (+ U DMA issue), F finish(), '
      'W DMA wait, B workgroup barrier')


def units_of(g, wv):
    return int((tr[g, wv, 2:] > 0).sum()) // NS


for g in (0, 1, G // 2, G - 1):
    print(f'-- workgroup {g}: start {a[g, :, 0].min():.0f}, prologue done at +{(a[g, :, 1] - a[g, :, 0]).max():.0f}')
    for wv in range(8):
        row = a[g, wv]
        segs = []
        prev = row[1]
        for k in range(min(units_of(g, wv), 6)):
            st = row[2 + NS * k: 2 + NS * (k + 1)]
            d = np.diff(np.concatenate([[prev], st]))
            segs.append(' '.join(f'{n}{v:5.0f}' for n, v in zip(names, d)))
            prev = st[-1]
        print(f'   wave {wv}: ' + ' | '.join(segs))
cols = {n: [] for n in names}
tot = []
for g in range(G):
    for wv in range(8):
        row = a[g, wv]
        for k in range(1, units_of(g, wv)):
            st = row[2 + NS * k: 2 + NS * (k + 1)]
            prev = row[2 + NS * k - 1]
            d = np.diff(np.concatenate([[prev], st]))
            for n, v in zip(names, d):
                cols[n].append(v)
            tot.append(st[-1] - prev)
print('# medians (means) over units >= 1 of all waves: ' + '  '.join(f'{n} {np.median(v):.0f} ({np.mean(v):.0f})' for n, v in cols.items())
      + f'  unit total {np.median(tot):.0f} ({np.mean(tot):.0f}); matrix work per unit and SIMD = 4096 cycles')
for grp_name, wvs in (('waves 0-3', range(4)), ('waves 4-7', range(4, 8))):
    sel = {n: [] for n in names}
    for g in range(G):
        for wv in wvs:
            row = a[g, wv]
            for k in range(1, units_of(g, wv)):
                st = row[2 + NS * k: 2 + NS * (k + 1)]
                d = np.diff(np.concatenate([[row[2 + NS * k - 1]], st]))
                for n, v in zip(names, d):
                    sel[n].append(v)
    print(f'#   {grp_name}: ' + '  '.join(f'{n} {np.median(v):.0f}' for n, v in sel.items()))
end = a.max(axis=(1, 2))
print(f'# launch: last stamp min / median / max over workgroups: {end.min():.0f} / {np.median(end):.0f} / {end.max():.0f} cycles')
