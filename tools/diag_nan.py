"""Where a long run on ONE minibatch ends: adapt(steps=1) on bench.py's default minibatch until the NaN guard fires
(dpp.py:1115-1118), printing the loss and the disparity range on the way -- to tell a collapsing disparity (the untrained
synthetic network trained hundreds of times on the same five triplets) from a kernel fault.

    python tools/diag_nan.py [max_steps=800]          (CLSLAM_NO_WINOGRAD=1 for the direct-convolution encoders)"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'cl-slam_amd'))

from bench import build_predictor  # noqa: E402
from clslam_hip import synth  # noqa: E402

H, W, B = 192, 640, 5
n_max = int(sys.argv[1]) if len(sys.argv) > 1 else 800
dev = torch.device('cuda:0')
torch.manual_seed(1)
p = build_predictor(H, W, B, host_outputs=False)
batch = {k: v.to(dev) for k, v in synth.make_batch(B, H, W, seed=0).items()}
print(f'winograd encoders: {os.environ.get("CLSLAM_NO_WINOGRAD", "0") != "1"}')
for it in range(n_max):
    try:
        out, losses = p.adapt(None, batch, steps=1)
    except RuntimeError as e:
        eng = p.engine
        eng.wait_training()
        w = eng.w
        print(f'step {it}: {e}; weights finite: {bool(torch.isfinite(w).all())}, max |w| {float(w.abs().max()):.3e}')
        ws = eng._ws[B]
        for s in range(4):
            d = ws.disp[s]
            print(f'  disp scale {s}: finite {bool(torch.isfinite(d).all())}  min {float(torch.nan_to_num(d).min()):.3e} '
                  f'max {float(torch.nan_to_num(d).max()):.3e}  exact zeros {int((d == 0).sum())} of {d.numel()}')
        break
    eng = p.engine
    eng.wait_training()
    if not bool(torch.isfinite(eng.w).all()) or not bool(torch.isfinite(eng.g).all()):
        import math
        print(f'step {it}: loss {float(losses["loss"]):.5f} is finite but the step left non-finite values behind:')
        for name, off, shape in eng.layout.entries:
            n = math.prod(shape)
            gb, wb = int((~torch.isfinite(eng.g[off:off + n])).sum()), int((~torch.isfinite(eng.w[off:off + n])).sum())
            if gb or wb:
                print(f'  {name:48s} {n:9d} values: gradient non-finite {gb}, weight non-finite {wb}')
        ws = eng._ws[B]
        print('  cam_T_cam[-1] translations:', ws.T[0, :, :3, 3].cpu().tolist())
        print('  cam_T_cam[+1] translations:', ws.T[1, :, :3, 3].cpu().tolist())
        for s in range(4):
            d = ws.disp[s]
            print(f'  disp scale {s}: min {float(d.min()):.6f} max {float(d.max()):.6f}')
        def scan(name, v):
            if torch.is_tensor(v) and v.is_floating_point() and v.numel():
                bad = int((~torch.isfinite(v)).sum())
                vv = torch.nan_to_num(v.double(), nan=0.0, posinf=0.0, neginf=0.0)
                print(f'  ws.{name:24s} {tuple(v.shape)!s:28s} non-finite {bad:9d}  finite range [{float(vv.min()):.4e}, {float(vv.max()):.4e}]')
            elif isinstance(v, (list, tuple)):
                for i, x in enumerate(v):
                    scan(f'{name}[{i}]', x)
            elif isinstance(v, dict):
                for i, x in v.items():
                    scan(f'{name}[{i}]', x)
        for name, v in sorted(vars(ws).items()):
            scan(name, v)
        for k in sorted(losses):
            print(f'  {k}: {float(losses[k]):.6f}')
        break
    if it % 25 == 0 or it > n_max - 3:
        d = out['disp', 0]
        print(f'step {it:4d}: loss {float(losses["loss"]):.5f}  disp0 min {float(d.min()):.3e} max {float(d.max()):.3e} '
              f'mean {float(d.mean()):.3e}')
else:
    print(f'no NaN in {n_max} steps')
