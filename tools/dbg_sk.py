import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'cl-slam_amd'))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'tests'))
from clslam_hip import _lib
if os.environ.get('CLSLAM_LIB'):
    _lib.LIB_PATH = Path(os.environ['CLSLAM_LIB']).resolve()
from clslam_hip import ops
from test_conv import _ref_conv
dev = torch.device('cuda:0')
torch.manual_seed(0)
ws = torch.zeros(16 << 20, dtype=torch.uint8, device=dev)
for (B, H, W, Ca, Cout) in ((1, 8, 16, 64, 32), (1, 16, 32, 64, 64)):
    xa = torch.randn(B, H, W, Ca)
    w = torch.randn(Cout, 9, Ca) * 0.1
    ref = _ref_conv(xa, w, ksize=3, stride=1, pad=1, pad_mode=0)
    for cfg in (30, 34, 35):
        for grp in (2, 3):
            os.environ['CLSLAM_SK_GROUPS'] = str(grp)
            out = torch.full((B, H, W, Cout), float('nan'), device=dev)
            ops.conv2d(xa.to(dev), w.to(dev), out, ksize=3, config=cfg, workspace=ws)
            o = out.cpu()
            err = (o - ref).abs()
            bad = err > 1e-4
            print(f'{H}x{W} Ca={Ca} Cout={Cout} cfg {cfg} G={grp}: max err {float(err.max()):.3e} bad {int(bad.sum())}/{bad.numel()}  nan {int(torch.isnan(o).sum())} flags {int(ws[:65536].view(torch.int32).abs().sum())}')
            if bad.any():
                idx = bad.nonzero()
                print('  bad rows(y):', sorted(set(idx[:, 1].tolist()))[:20], ' x:', sorted(set(idx[:, 2].tolist()))[:40], ' ch:', sorted(set(idx[:, 3].tolist()))[:40])
                y, x, c = [int(v) for v in idx[0, 1:]]
                print('  first bad', (y, x, c), float(o[0, y, x, c]), float(ref[0, y, x, c]))
