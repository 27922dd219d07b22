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
    l['loss'].backward(retain_graph=True)
    full = outs['depth', sc].grad.clone()      # (retain_grad's hook would add the partial passes below to .grad)
    wgrads = {f: outs['rgb', f, sc].grad.clone() for f in (-1, 1)}
    parts = {}
    for f in (-1, 1):       # dL/d depth through ONE source frame's synthesised image only
        parts[f] = torch.autograd.grad(outs['rgb', f, sc], outs['depth', sc], grad_outputs=wgrads[f], retain_graph=True)[0][:, 0]
    outs['depth', sc].grad = full
    for f in (-1, 1):
        outs['rgb', f, sc].grad = wgrads[f]
    return o, outs, parts


o, outs, parts = run_depth()
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
print('  exact dL/d depth there through frame -1 only:', float(parts[-1][bw, y, x]), ' through frame +1 only:', float(parts[1][bw, y, x]),
      ' sum', float(parts[-1][bw, y, x] + parts[1][bw, y, x]))
# the same split for every pixel: where does the kernel path equal ONE part instead of the sum?
only_m1 = ((mine - parts[-1]).abs() < 1e-3 * ex.abs().clamp_min(1e-12)) & (parts[1].abs() > 1e-2 * ex.abs().clamp_min(1e-12))
only_p1 = ((mine - parts[1]).abs() < 1e-3 * ex.abs().clamp_min(1e-12)) & (parts[-1].abs() > 1e-2 * ex.abs().clamp_min(1e-12))
bad = (mine - ex).abs() > 0.05 * ex.abs().clamp_min(1e-3 * float(ex.abs().max()))
print(f'  pixels off by > 5 %: {int(bad.sum())}; of all pixels, kernel value == frame -1 part only: {int(only_m1.sum())}, == frame +1 part only: {int(only_p1.sum())}')
ys, xs = torch.nonzero(bad.any(0), as_tuple=True)
print('  rows of the bad pixels mod 8:', sorted(set((ys % 8).tolist())), ' list:', [(int(b_), int(y_), int(x_)) for b_, y_, x_ in torch.nonzero(bad)[:12]])

# ---- the SSIM derivative coefficients the forward stored for the selected frame (ws.coef, planes c*3+j = alpha/beta/gamma of
# channel c) against a float64 recomputation from the kernel path's OWN synthesised images: d(0.85/3 * clamp((1-S)/2))/dx_p of
# pixel q's SSIM w.r.t. any pixel p of its 3x3 patch = alpha_c + beta_c x_p + gamma_c y_p
import torch.nn.functional as F
coefs = ws.coef[sc].cpu().double()                       # (B, 9, H, W)
selc = sel[sc].long()
tgt = batch['rgb', 0, 0].double()
ref = torch.zeros_like(coefs)
C1, C2 = 0.01 ** 2, 0.03 ** 2
for fi, f in enumerate((-1, 1)):
    xw = out['rgb', f, sc].detach().cpu().double()
    xp, yp = F.pad(xw, (1, 1, 1, 1), mode='reflect'), F.pad(tgt, (1, 1, 1, 1), mode='reflect')
    mu_x, mu_y = F.avg_pool2d(xp, 3, 1), F.avg_pool2d(yp, 3, 1)
    sx, sy = F.avg_pool2d(xp * xp, 3, 1) - mu_x ** 2, F.avg_pool2d(yp * yp, 3, 1) - mu_y ** 2
    sxy = F.avg_pool2d(xp * yp, 3, 1) - mu_x * mu_y
    n1, n2 = 2 * mu_x * mu_y + C1, 2 * sxy + C2
    d1, d2 = mu_x ** 2 + mu_y ** 2 + C1, sx + sy + C2
    S = n1 * n2 / (d1 * d2)
    raw = (1 - S) / 2
    kf = torch.where((raw >= 0) & (raw <= 1), torch.full_like(raw, (0.85 / 3) * (-0.5) / 9), torch.zeros_like(raw)) / (d1 * d2)
    al, be, ga = kf * (2 * mu_y * (n2 - n1) - S * 2 * mu_x * (d2 - d1)), kf * (-2 * S * d1), kf * (2 * n1)
    mine_f = torch.stack([al[:, 0], be[:, 0], ga[:, 0], al[:, 1], be[:, 1], ga[:, 1], al[:, 2], be[:, 2], ga[:, 2]], 1)
    m = (selc == 2 + fi).unsqueeze(1)
    ref = torch.where(m, mine_f, ref)
used = (selc >= 2).unsqueeze(1).expand_as(coefs)
err = ((coefs - ref).abs() * used)
scale = ref.abs().amax(1, keepdim=True).clamp_min(1e-9)
relc = (err / scale).amax(1)
print(f'stored SSIM coefficients vs float64 recomputation (scale {sc}): worst relative error {float(relc.max()):.2e}; pixels above 1e-2: {int((relc > 1e-2).sum())}, '
      f'above 1e-3: {int((relc > 1e-3).sum())} of {int((selc >= 2).sum())}')
for b_, y_, x_ in torch.nonzero(relc > 1e-2)[:8]:
    print(f'   sample {int(b_)} pixel ({int(y_)},{int(x_)}) sel {int(selc[b_, y_, x_])}: stored {coefs[b_, :3, y_, x_].tolist()} float64 {ref[b_, :3, y_, x_].tolist()}')

# ---- at the worst pixel: dL/d(synthesised image) rebuilt in float64 from the STORED coefficients (what the backward kernel
# combines: sum over the 3x3 neighbours q that selected the frame of alpha_q + beta_q x_p + gamma_q y_p, + the L1 sign if p itself did)
wq = (1.0 / B) / (H * W) / 4.0
for fi, f in enumerate((-1, 1)):
    xw = out['rgb', f, sc].detach().cpu().double()[bw, :, y, x]
    yv = tgt[bw, :, y, x]
    g = torch.zeros(3, dtype=torch.float64)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            qy, qx = y + dy, x + dx
            if not (0 <= qy < H and 0 <= qx < W) or int(selc[bw, qy, qx]) != 2 + fi:
                continue
            wgt = (2.0 if (qy == 0 and y == 1) or (qy == H - 1 and y == H - 2) else 1.0) * (2.0 if (qx == 0 and x == 1) or (qx == W - 1 and x == W - 2) else 1.0)
            cq = coefs[bw, :, qy, qx].reshape(3, 3)          # [channel][alpha, beta, gamma]
            g += wgt * (cq[:, 0] + cq[:, 1] * xw + cq[:, 2] * yv)
    if int(selc[bw, y, x]) == 2 + fi:
        g += (0.15 / 3) * torch.sign(xw - yv)
    print(f'  frame {f}: dL/d warped from the stored coefficients {(g * wq).tolist()}   exact {outs["rgb", f, sc].grad[bw, :, y, x].tolist()}')

# ---- the geometric part at the worst pixel in float64 from the kernel path's own inputs (projection matrices, inverse
# intrinsics, depth, source image): does the FORMULA give the oracle's number or the kernel's?
Pall = ws.P.cpu().double()                                  # (2, B, 3, 4)
Kinv = ws.ctx.Kinv.cpu().double()[bw]
dpx = float(dep[bw, y, x])
cam = Kinv[:3, :3] @ torch.tensor([float(x), float(y), 1.0], dtype=torch.float64)
tot = 0.0
for fi, f in enumerate((-1, 1)):
    Pm = Pall[fi, bw]
    pvec = Pm[:, :3] @ (dpx * cam) + Pm[:, 3]
    den = pvec[2] + 1e-7
    u, v = pvec[0] / den, pvec[1] / den
    ix, iy = min(max(float(u), 0.0), W - 1.0), min(max(float(v), 0.0), H - 1.0)
    mx, my = float(0.0 < u < W - 1), float(0.0 < v < H - 1)
    c = cells[sc, fi, bw, y, x].item()
    cx0, cy0 = c & 0xFFF, (c >> 12) & 0xFFF
    src = batch['rgb', f, 0][bw].double()
    x1, y1 = min(cx0 + 1, W - 1), min(cy0 + 1, H - 1)
    x1ok, y1ok = float(cx0 + 1 <= W - 1), float(cy0 + 1 <= H - 1)
    wx1, wy1 = ix - cx0, iy - cy0
    wx0, wy0 = 1 - wx1, 1 - wy1
    g = outs['rgb', f, sc].grad[bw, :, y, x]
    nw, ne, sw, se = src[:, cy0, cx0], src[:, cy0, x1] * x1ok, src[:, y1, cx0] * y1ok, src[:, y1, x1] * x1ok * y1ok
    gix = float((g * (-nw * wy0 + ne * wy0 - sw * wy1 + se * wy1)).sum()) * mx
    giy = float((g * (-nw * wx0 - ne * wx1 + sw * wx0 + se * wx1)).sum()) * my
    a = Pm[:, :3] @ cam
    dd = (gix * (a[0] * (Pm[2, 3] + 1e-7) - a[2] * Pm[0, 3]) + giy * (a[1] * (Pm[2, 3] + 1e-7) - a[2] * Pm[1, 3])) / den ** 2
    tot += float(dd)
    print(f'  frame {f}: float64 formula: u {float(u):.5f} v {float(v):.5f} den {float(den):.6f} cell ({cx0},{cy0}) floor ({int(ix)},{int(iy)}) '
          f'unclipped x/y {mx:.0f}/{my:.0f} (kernel flags {(c >> 24) & 1}/{(c >> 25) & 1}) du {gix:.4e} dv {giy:.4e} -> dL/d depth {float(dd):.6e}')
print(f'  float64 formula total {tot:.6e}   kernel path {float(mine[bw, y, x]):.6e}   oracle {float(ex[bw, y, x]):.6e}')
