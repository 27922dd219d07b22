"""Diagnostic: where does the POSE gradient of the kernel path leave the exact one?  The pose path of the
backward is  dL/d(warped image) -> [loss_bwd2: grid_sample / projection backward, summed over pixels] -> dL/dP (24 numbers per
sample) -> [pose_bwd: K^T, Rodrigues chain, velocity term] -> dL/d(axis-angle, translation) -> pose decoder.  This takes the
step apart at dL/dP:
  (1) the kernel path's dL/dP (its block partials, summed here in double) against float64 autograd of the same view synthesis
      at the SAME forward point (kernel path's disparities, projection matrices, cells, clip flags) with the oracle's exact
      dL/d(warped image);
  (2) the kernel path's dL/d(pose) against the float64 chain applied to ITS OWN dL/dP (the arithmetic of pose_bwd alone);
  (3) torch's fp32 autograd at the same point, for scale.
    python tools/diag_pose.py hip 192 640 5 5"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
    sys.path.insert(0, str(p))
import torch
import torch.nn.functional as F
from clslam_hip import ops, synth
from emu_util import use_backend
from helpers import make_oracle, rel_l2
from oracle import functional as OF
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
         **{('cam_T_cam', 0, f): out['cam_T_cam', 0, f].detach().cpu() for f in (-1, 1)},
         **{('P', f): ws.P[fi].detach().cpu() for fi, f in enumerate((-1, 1))}}
l1_sign = {s: {f: torch.sign(batch['rgb', 0, 0] - out['rgb', f, s].detach().cpu()) for f in (-1, 1)} for s in range(4)}


def run(double):
    o = make_oracle(H, W, B)
    if double:
        o.to_double()
    o.forced_sel = {s: sel[s] for s in range(4)}
    o.forced_cells, o.forced_forward = fc, point
    o.forced_l1_sign = {s: {f: (v.double() if double else v) for f, v in d.items()} for s, d in l1_sign.items()}
    b = {k: (v.double() if (double and v.is_floating_point()) else v) for k, v in batch.items()}
    n = {s: (v.double() if double else v) for s, v in noise.items()}
    o.set_adapt()
    outs, l = o.process_batch(b, n, None)
    for s in range(4):
        for f in (-1, 1):
            outs['rgb', f, s].retain_grad()
    for f in (-1, 1):
        outs['axis_angle', 0, f].retain_grad()
        outs['translation', 0, f].retain_grad()
    o.optimizer.zero_grad()
    l['loss'].backward()
    return outs


o64, o32 = run(True), run(False)
dt_ = torch.float64
Kinv = batch['inv_camera_matrix', 0].to(dt_)
K = batch['camera_matrix', 0].to(dt_)


def exact_dP(outs, dtype):
    """autograd of the view synthesis w.r.t. the projection matrices at the kernel path's forward point, fed with `outs`'
    dL/d(warped image); -> (2, B, 3, 4)"""
    tot = torch.zeros(2, B, 3, 4, dtype=dtype)
    for s in range(4):
        disp = F.interpolate(point['disp', s].to(dtype), [H, W], mode='bilinear', align_corners=False)
        depth = OF.disp_to_depth(disp, p.min_depth, p.max_depth)
        pts = OF.backproject(depth, batch['inv_camera_matrix', 0].to(dtype))
        for fi, f in enumerate((-1, 1)):
            P = ws.P[fi].detach().cpu().to(dtype).clone().requires_grad_(True)
            cam = torch.matmul(P, pts)
            pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + 1e-7)
            pix = pix.view(B, 2, H, W).permute(0, 2, 3, 1)
            grid = (torch.stack([pix[..., 0] / (W - 1), pix[..., 1] / (H - 1)], -1) - 0.5) * 2
            warped = OF.grid_sample_border(batch['rgb', f, 0].to(dtype), grid, fc[s][f])
            tot[fi] += torch.autograd.grad(warped, P, grad_outputs=outs['rgb', f, s].grad.to(dtype))[0]
    return tot


dP_exact = exact_dP(o64, torch.float64)
dP_t32 = exact_dP(o32, torch.float32).double()
dP_k = t.dp_partial.cpu().sum((0, 2)).reshape(B, 2, 3, 4).permute(1, 0, 2, 3)          # block partials (double) -> (2, B, 3, 4)
print(f'dL/dP ({H}x{W}, B={B}): relative L2 vs float64 autograd at the same forward point, per frame and sample')
for fi, f in enumerate((-1, 1)):
    print(f'  frame {f:+d}: kernels', ' '.join(f'{rel_l2(dP_k[fi, b], dP_exact[fi, b]):.1e}' for b in range(B)),
          '  torch fp32', ' '.join(f'{rel_l2(dP_t32[fi, b], dP_exact[fi, b]):.1e}' for b in range(B)))
print('  entry by entry, worst sample of frame -1 (row-major 3x4; kernel / exact):')
bw = max(range(B), key=lambda b: rel_l2(dP_k[0, b], dP_exact[0, b]))
for i in range(3):
    print('     ', ' '.join(f'{float(dP_k[0, bw, i, j]): .6e}/{float(dP_exact[0, bw, i, j]): .6e}' for j in range(4)))
# K^T dP: the combination the pose chain actually sees (rows of K^T mix dP rows with weights fx, fy, cx, cy ~ 300-600)
for name, dP in (('kernels', dP_k), ('torch fp32', dP_t32)):
    dM = torch.einsum('bik,fbij->fbkj', K[:, :3, :], dP)
    dMe = torch.einsum('bik,fbij->fbkj', K[:, :3, :], dP_exact)
    print(f'  K^T dL/dP, {name}: per frame and sample', ' '.join(f'{rel_l2(dM[fi, b], dMe[fi, b]):.1e}' for fi in range(2) for b in range(B)))


def chain(dP):
    """float64 pose chain (utils.py:34-117, layers.py:94) + velocity term applied to a given dL/dP -> (2, B, 6)"""
    pose = ws.pose.detach().cpu().double()
    res = torch.zeros(2, B, 6, dtype=torch.float64)
    aa = {f: pose[fi * B:(fi + 1) * B, 0:3].reshape(B, 1, 3).clone().requires_grad_(True) for fi, f in enumerate((-1, 1))}
    tr = {f: pose[fi * B:(fi + 1) * B, 3:6].reshape(B, 1, 3).clone().requires_grad_(True) for fi, f in enumerate((-1, 1))}
    L = 0.0
    for fi, f in enumerate((-1, 1)):
        T = OF.transformation_from_parameters(aa[f], tr[f], invert=f < 0)
        P = torch.matmul(K, T)[:, :3, :]
        L = L + (P * dP[fi]).sum()
    v = 0.05 * OF.velocity_loss(tr[-1], tr[1], batch['relative_distance', 0], batch['relative_distance', 1])
    L = L + (v * (torch.ones(B, dtype=torch.float64) / B)).sum()
    L.backward()
    for fi, f in enumerate((-1, 1)):
        res[fi, :, :3], res[fi, :, 3:] = aa[f].grad.reshape(B, 3), tr[f].grad.reshape(B, 3)
    return res


mine = torch.stack([t.dpose[fi * B:(fi + 1) * B, :6].cpu().double() for fi in range(2)])
ex = torch.stack([torch.cat([o64['axis_angle', 0, f].grad.reshape(B, 3), o64['translation', 0, f].grad.reshape(B, 3)], 1) for f in (-1, 1)])
t32 = torch.stack([torch.cat([o32['axis_angle', 0, f].grad.reshape(B, 3), o32['translation', 0, f].grad.reshape(B, 3)], 1) for f in (-1, 1)]).double()
own = chain(dP_k)
chk = chain(dP_exact)
print('dL/d(axis-angle, translation): relative L2 per frame and sample')
print('  kernels vs float64 oracle        ', ' '.join(f'{rel_l2(mine[fi, b], ex[fi, b]):.1e}' for fi in range(2) for b in range(B)))
print('  torch fp32 vs float64 oracle     ', ' '.join(f'{rel_l2(t32[fi, b], ex[fi, b]):.1e}' for fi in range(2) for b in range(B)))
print('  kernels vs float64 chain on their OWN dL/dP (pose_bwd arithmetic alone)', ' '.join(f'{rel_l2(mine[fi, b], own[fi, b]):.1e}' for fi in range(2) for b in range(B)))
print('  float64 chain on the exact dL/dP vs float64 oracle (consistency of this tool)', ' '.join(f'{rel_l2(chk[fi, b], ex[fi, b]):.1e}' for fi in range(2) for b in range(B)))
print('  whole (2, B, 6): kernels', f'{rel_l2(mine, ex):.2e}', ' torch fp32', f'{rel_l2(t32, ex):.2e}')
