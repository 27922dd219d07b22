#!/usr/bin/env python
"""Time the non-conv kernels of the adapt step one by one on the real 192x640 B=5 workspace."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'cl-slam_amd')); sys.path.insert(0, str(ROOT))
import torch
import bench
from clslam_hip import ops, synth

H, W, B = 192, 640, 5
p = bench.build_predictor(H, W, B)
eng = p.engine
batch = {k: v.to(p.device) for k, v in synth.make_batch(B, H, W, seed=0).items()}
for _ in range(2):
    p.adapt(None, batch, steps=1)
torch.cuda.synchronize()
ws = eng._ws[B]


def timeit(name, fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:34s} {e0.elapsed_time(e1) / iters * 1e3:8.1f} us', flush=True)


aug = {f: batch['rgb_aug', f, 0].contiguous() for f in (-1, 0, 1)}
e = eng.enc['depth_encoder']
pe = eng.enc['pose_encoder']
timeit('stem depth (3ch, B)', lambda: ops.stem_conv(aug[0], None, e.stem_w, e.stem_scale, e.stem_shift, ws.denc.f0))
timeit('stem pose (6ch, B)', lambda: ops.stem_conv(aug[-1], aug[0], pe.stem_w, pe.stem_scale, pe.stem_shift, ws.penc.f0[:B]))
timeit('maxpool depth', lambda: ops.maxpool3x3s2(ws.denc.f0, ws.denc.pool))
timeit('maxpool pose (2B)', lambda: ops.maxpool3x3s2(ws.penc.f0, ws.penc.pool))
from clslam_hip.engine import NUM_CH_DEC, ACT_ELU
t = ws.train
for i in range(4):
    hi, wi, ci = H >> i, W >> i, NUM_CH_DEC[i]
    wd, bd = eng._wb(f'depth_decoder/dispconv_{i}.conv', 1, ci, 9)
    timeit(f'dispconv_fwd s{i} ({hi}x{wi}x{ci})', lambda: ops.dispconv_fwd(ws.x[i, 1], wd.view(9, ci), bd, ws.disp[i]))
    dxp = t.dxp[0][:B * (hi + 2) * (wi + 2) * ci].view(B, hi + 2, wi + 2, ci)
    timeit(f'dispconv_bwd_data s{i}', lambda: ops.dispconv_bwd_data(t.dz_disp[i], wd.view(9, ci), dxp, ci, accumulate=True))
    timeit(f'dispconv_wgrad s{i}', lambda: ops.dispconv_wgrad(t.dz_disp[i], ws.x[i, 1], t.disp_part[i]))
    timeit(f"fold_act_grad s{i} nopool+disp", lambda: ops.fold_act_grad(dxp, ws.x[i, 1], t.dz[i, 1], h=hi, w=wi, ch=ci, border=1, pool=False, disp_dz=t.dz_disp[i], disp_w=wd.view(9, ci),
                                                                   act=ACT_ELU, bias_partial=t.bias_part[i, 1]))
    timeit(f'fold_act_grad s{i} pool', lambda: ops.fold_act_grad(dxp, ws.x[i, 0], t.dz[i, 0], h=hi, w=wi, ch=ci, border=1, pool=True,
                                                                 act=ACT_ELU, bias_partial=t.bias_part[i, 0]))
c = ws.ctx
timeit('warp_fwd_pyramid', lambda: ops.warp_fwd_pyramid(ws.disp, c.rgb[-1], c.rgb[1], c.Kinv, ws.P, ws.depth, ws.warped, eng.min_depth, eng.max_depth))
timeit('photo_automask_pyramid', lambda: ops.photo_automask_pyramid(ws.warped, c.rgb[0], ws.idmap, ws.noise, ws.sel, ws.coef, ws.partial, B, H, W))
timeit('loss_bwd2_pyramid', lambda: ops.loss_bwd2_pyramid(ws.disp, ws.sel, ws.coef, ws.warped, c.rgb[0], c.rgb[-1], c.rgb[1], c.Kinv, ws.P, c.sample_w,
                                                          t.ddisp_up, t.dp_partial, eng.min_depth, eng.max_depth))
timeit('disp_grad_pyramid', lambda: ops.disp_grad_pyramid(t.ddisp_up, ws.disp, c.aux if c.n_smooth else None, c.n_smooth, t.dz_disp, H, W))
timeit('pose_bwd', lambda: ops.pose_bwd(t.dp_partial, 4, t.nb2, ws.pose, c.K, c.d0, c.d1, c.sample_w, eng.vel_scale, t.dpose))
w2_, b2_ = eng._wb('pose_decoder/pose_2', 12, 256, 1)
timeit('pose_head_fwd', lambda: ops.pose_head_fwd(ws.p1, w2_.view(12, 256), b2_, ws.pmean, ws.pose))
timeit('adam', lambda: ops.adam_step(eng.w, eng.g, eng.m, eng.v, 1e-4, 3))
