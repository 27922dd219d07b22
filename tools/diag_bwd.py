"""Diagnostic: where does the backward of the kernel path differ from the exact (float64 oracle) backward evaluated at the same
decisions and the same forward point?  Prints per-scale / per-sample errors of dL/d(disp_s) and of dL/d(pose), kernels and
torch fp32 side by side.   python tools/diag_bwd.py [hip|emu] H W B seed"""
import math
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
    sys.path.insert(0, str(p))
import torch
from clslam_hip import ops, synth
from emu_util import use_backend
from helpers import make_oracle, rel_l2
from predictor_util import make_predictor

backend = sys.argv[1] if len(sys.argv) > 1 else 'emu'
H, W, B, seed = (int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (64, 128, 2, 3)))
dev = use_backend(backend)
p = make_predictor(H, W, B)
batch = synth.make_batch(B, H, W, seed=seed)
noise = synth.make_noise(B, H, W, seed=seed + 10)
p.set_tie_break_noise(noise)
out, losses = p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
eng = p.engine
eng.wait_training()
ws = eng._ws[B]
t = ws.train
sel = ws.sel.cpu().clone()
cells = torch.empty(4, 2, B, H, W, dtype=torch.int32, device=dev)
ops.warp_cells_pyramid(ws.disp, ws.ctx.Kinv, ws.P, cells, p.min_depth, p.max_depth)
cells = cells.cpu()
fc = {}
for s in range(4):
    fc[s] = {}
    for fi, f in enumerate((-1, 1)):
        c = cells[s, fi].long()
        fc[s][f] = ((c & 0xFFF), ((c >> 12) & 0xFFF), ((c >> 24) & 1).bool(), ((c >> 25) & 1).bool())
point = {**{('disp', s): out['disp', s].detach().cpu() for s in range(4)},
         **{('cam_T_cam', 0, f): out['cam_T_cam', 0, f].detach().cpu() for f in (-1, 1)}}


l1_sign = {s: {f: torch.sign(batch['rgb', 0, 0] - out['rgb', f, s].detach().cpu()) for f in (-1, 1)} for s in range(4)}


def run(double):
    o = make_oracle(H, W, B)
    if double:
        for m in o.models.values():
            m.double()
    o.forced_sel = {s: sel[s] for s in range(4)}
    o.forced_cells, o.forced_forward = fc, point
    if 'nosign' not in sys.argv:
        o.forced_l1_sign = {s: {f: (v.double() if double else v) for f, v in d.items()} for s, d in l1_sign.items()}
    b = {k: (v.double() if (double and v.is_floating_point()) else v) for k, v in batch.items()}
    n = {s: (v.double() if double else v) for s, v in noise.items()}
    o.set_adapt()
    outs, l = o.process_batch(b, n, None)
    for s in range(4):
        outs['disp', s].retain_grad()
    for f in (-1, 1):
        outs['axis_angle', 0, f].retain_grad()
        outs['translation', 0, f].retain_grad()
    o.optimizer.zero_grad()
    l['loss'].backward()
    return outs


o64, o32 = run(True), run(False)
for s in range(4):
    d = ws.disp[s].cpu()
    mine = (t.dz_disp[s].cpu() / (d * (1 - d))).double()
    ex, t32 = o64['disp', s].grad[:, 0], o32['disp', s].grad[:, 0].double()
    print(f'dL/d disp scale {s}: kernels {rel_l2(mine, ex):.2e}  torch32 {rel_l2(t32, ex):.2e}   per sample kernels',
          ' '.join(f'{rel_l2(mine[b], ex[b]):.1e}' for b in range(B)), ' torch32', ' '.join(f'{rel_l2(t32[b], ex[b]):.1e}' for b in range(B)))
    e = (mine - ex).abs()
    for b in range(B):
        i = int(e[b].argmax())
        print(f'      sample {b}: worst pixel ({i // ex.shape[-1]},{i % ex.shape[-1]}) err {float(e[b].max()):.2e} vs |g|max {float(ex[b].abs().max()):.2e} '
              f'rms {float(ex[b].pow(2).mean().sqrt()):.2e}; disp there {float(d[b].reshape(-1)[i]):.3e}')
for fi, f in enumerate((-1, 1)):
    mine = t.dpose[fi * B:(fi + 1) * B].cpu().double()
    r = torch.cat([o64['axis_angle', 0, f].grad.reshape(B, 3), o64['translation', 0, f].grad.reshape(B, 3)], 1)
    r32 = torch.cat([o32['axis_angle', 0, f].grad.reshape(B, 3), o32['translation', 0, f].grad.reshape(B, 3)], 1).double()
    print(f'dL/d pose frame {f}: per sample kernels', ' '.join(f'{rel_l2(mine[b, :6], r[b]):.1e}' for b in range(B)),
          ' torch32', ' '.join(f'{rel_l2(r32[b], r[b]):.1e}' for b in range(B)))

# ---- the single worst high-resolution sample of dL/d depth (scale given by argv[6], default 0): everything about it ----
sc = int(sys.argv[6]) if len(sys.argv) > 6 and sys.argv[6].isdigit() else 0
dep = out['depth', sc].detach().cpu()[:, 0].double()
d_up = 0.1 / dep                                   # min_depth / depth = upsampled disparity (adapt config)
mine = t.ddisp_up[sc].cpu().double() / (-dep / d_up)          # dL/d depth from dL/d(upsampled disp)


def depth_grad(o):
    return None


# re-run the float64 oracle keeping dL/d depth
def run_depth():
    o = make_oracle(H, W, B)
    for m in o.models.values():
        m.double()
    o.forced_sel = {s: sel[s] for s in range(4)}
    o.forced_cells, o.forced_forward = fc, point
    o.forced_l1_sign = {s: {f: v.double() for f, v in d.items()} for s, d in l1_sign.items()}
    o.record_cells = False
    b = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    n = {s: v.double() for s, v in noise.items()}
    o.set_adapt()
    outs, l = o.process_batch(b, n, None)
    outs['depth', sc].retain_grad()
    for f in (-1, 1):
        outs['rgb', f, sc].retain_grad()
    l['loss'].backward()
    return o, outs


o, outs = run_depth()
ex = outs['depth', sc].grad[:, 0]
e = (mine - ex).abs()
bw = int(e.reshape(B, -1).max(1).values.argmax())
i = int(e[bw].argmax())
y, x = i // W, i % W
print(f'worst dL/d depth at scale {sc}: sample {bw} pixel ({y},{x}): kernels {float(mine[bw, y, x]):.6e} exact {float(ex[bw, y, x]):.6e}; '
      f'rms of the map {float(ex[bw].pow(2).mean().sqrt()):.3e}')
print('  selection 3x3 around it (0,1: identity; 2,3: frames -1,+1):', sel[sc, bw, max(0, y - 1):y + 2, max(0, x - 1):x + 2].tolist())
for fi, f in enumerate((-1, 1)):
    c = cells[sc, fi, bw, y, x].item()
    wv = out['rgb', f, sc].detach().cpu()[bw, :, y, x]
    tv = batch['rgb', 0, 0][bw, :, y, x]
    print(f'  frame {f}: cell x0={c & 0xFFF} y0={(c >> 12) & 0xFFF} not-clipped x={(c >> 24) & 1} y={(c >> 25) & 1}; warped {wv.tolist()} target {tv.tolist()} '
          f'oracle warped {outs["rgb", f, sc].detach()[bw, :, y, x].tolist()}')
    gw = outs['rgb', f, sc].grad[bw, :, y, x]
    print(f'           exact dL/d warped there {gw.tolist()}')
print('  depth', float(dep[bw, y, x]), 'disp_up', float(d_up[bw, y, x]))
nb = e[bw, max(0, y - 2):y + 3, max(0, x - 2):x + 3]
print('  |err| 5x5 neighbourhood / rms:', (nb / ex[bw].pow(2).mean().sqrt()).numpy().round(3).tolist())
