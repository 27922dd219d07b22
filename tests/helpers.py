"""Shared helpers for the parity tests (oracle construction, golden loading, error metrics)."""
from pathlib import Path

import numpy as np
import torch

from clslam_hip import synth

GOLDEN = Path(__file__).resolve().parent / 'golden'


def load_golden(name: str):
    return dict(np.load(GOLDEN / f'{name}.npz', allow_pickle=False))


def make_oracle(H: int, W: int, B: int, seed: int = 0, **kw):
    from oracle import OraclePredictor
    p = OraclePredictor(H, W, B, **kw)
    for name, m in p.models.items():
        m.load_state_dict(synth.fill_state_dict(m.state_dict(), seed, name))
    return p


def rel_err(a, b) -> float:
    """max |a-b| / max(|b|) -- the relative error used for the 1e-4 parity bar."""
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def max_rel(a, b, floor: float = 1e-6) -> float:
    """elementwise max |a-b| / max(|b|, floor)."""
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float(((a - b).abs() / b.abs().clamp_min(floor)).max())


# ---- attributed backward parity (tests/test_backward_parity.py, tests/test_predictor.py) ---------------------------
def oracle_grads(o, batch, noise):
    o.set_adapt()
    out, losses = o.process_batch(batch, noise, None)
    o.optimizer.zero_grad()
    losses['loss'].backward()
    grads = {}
    for model in ('depth_decoder', 'pose_decoder'):
        for k, prm in o.models[model].named_parameters():
            grads[f'{model}/{k}'] = prm.grad.detach().clone()
    return out, {k: v.detach() for k, v in losses.items()}, grads


def rel_l2(a, b) -> float:
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def attributed_gradient_errors(p, batch, noise, dev):
    """`p`: the product predictor right after ONE adapt(steps=1) on (batch, noise).  Compares its 36 gradient tensors (full
    tensors, relative L2) with the oracle's autograd three times: free; with the oracle forced to the kernel path's 4-way-min
    selection; and with the kernel path's bilinear cells and border-clip flags imposed on the oracle's written-out sampler
    as well (clslam_warp_cells_pyramid -> oracle.functional.grid_sample_border).  Returns a dict with the flip counts,
    the rows (name, e_free, e_same_selection, norm, e_same_decisions, kernels vs float64 oracle on those decisions, torch fp32
    vs the same, kernels vs float64 at the kernel path's forward point, torch fp32 vs the same, and both once more with the
    kernel path's rounded projection matrices part of that point) and the free oracle's losses / gradients."""
    import math
    from clslam_hip import ops
    from clslam_hip.engine import TrainableLayout
    eng = p.engine
    B, H, W = batch['rgb', 0, 0].shape[0], p.height, p.width
    eng.wait_training()
    hip = {name: TrainableLayout.to_reference(eng.g[off:off + math.prod(shape)], shape).cpu().clone()
           for name, off, shape in eng.layout.entries}
    ws = eng._ws[B]
    sel_hip = ws.sel.cpu().clone()                       # (4, B, H, W) u8
    o = make_oracle(H, W, B)
    _, ol, ref = oracle_grads(o, batch, noise)
    # (a) selection flips: few, and all of them near-ties
    flips, worst_gap = 0, 0.0
    for s in range(4):
        so, sh = o.last_sel[s], sel_hip[s].long()
        diff = so != sh
        flips += int(diff.sum())
        if diff.any():
            comb = o.last_combined[s]
            a = torch.gather(comb, 1, so.unsqueeze(1)).squeeze(1)[diff]
            b = torch.gather(comb, 1, sh.unsqueeze(1)).squeeze(1)[diff]
            worst_gap = max(worst_gap, float((b - a).abs().max()))
    # (b) the oracle on the kernel path's selection
    o2 = make_oracle(H, W, B)
    o2.forced_sel = {s: sel_hip[s] for s in range(4)}
    o2.record_cells = True
    _, _, forced = oracle_grads(o2, batch, noise)
    # (c) ... and on the kernel path's bilinear cells / clip flags as well
    cells = torch.empty(4, 2, B, H, W, dtype=torch.int32, device=dev)
    ops.warp_cells_pyramid(ws.disp, ws.ctx.Kinv, ws.P, cells, p.min_depth, p.max_depth)
    cells = cells.cpu()
    forced_cells, cell_flips, clip_flips, far_cells = {}, 0, 0, 0
    for s in range(4):
        forced_cells[s] = {}
        for fi, f in enumerate((-1, 1)):
            c = cells[s, fi].long()
            mine = ((c & 0xFFF), ((c >> 12) & 0xFFF), ((c >> 24) & 1).bool(), ((c >> 25) & 1).bool())
            forced_cells[s][f] = mine
            theirs = o2.last_cells[s][f]
            cell_flips += int(((mine[0] != theirs[0]) | (mine[1] != theirs[1])).sum())
            # more than one cell apart: not a sample on a cell boundary but an ill-conditioned projection (denominator near 0)
            far_cells += int((((mine[0] - theirs[0]).abs() > 1) | ((mine[1] - theirs[1]).abs() > 1)).sum())
            clip_flips += int(((mine[2] != theirs[2]) | (mine[3] != theirs[3])).sum())
    o3 = make_oracle(H, W, B)
    o3.forced_sel = {s: sel_hip[s] for s in range(4)}
    o3.forced_cells = forced_cells
    _, _, forced3 = oracle_grads(o3, batch, noise)
    # (d) the same decisions once more in float64: what is left between an fp32 implementation and THIS is rounding.  The
    # gradients are sums of millions of terms of both signs (and 1/depth, 1/den factors of an untrained network), so fp32
    # rounding alone moves them by 1e-4 ... 1e-2 of their norm -- in torch's fp32 as much as in the kernels'.
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    n64 = {s: v.double() for s, v in noise.items()}

    # the kernel path's sign(target - synthesised image), the kink of the L1 term: imposed together with the forward point
    target = batch['rgb', 0, 0]
    outs = p.engine._outputs(ws, B)
    l1_sign = {s: {f: torch.sign(target - outs['rgb', f, s].detach().cpu()) for f in (-1, 1)} for s in range(4)}

    def exact_run(forward_point, double=True):
        o4 = make_oracle(H, W, B)
        if double:
            for m in o4.models.values():
                m.double()
        o4.forced_sel, o4.forced_cells, o4.forced_forward = o3.forced_sel, forced_cells, forward_point
        if forward_point is not None:
            o4.forced_l1_sign = {s: {f: v.double() if double else v for f, v in d.items()} for s, d in l1_sign.items()}
        got = oracle_grads(o4, b64 if double else batch, n64 if double else noise)
        if forward_point is not None and double:      # how many L1 signs the exact evaluation would have taken the other way
            nonlocal sign_flips
            sign_flips = sum(int(((torch.sign(b64['rgb', 0, 0] - got[0]['rgb', f, s].detach()) != l1_sign[s][f]) & (l1_sign[s][f] != 0)).sum())
                             for s in range(4) for f in (-1, 1))
        return got[2]
    sign_flips = 0
    exact = exact_run(None)
    # (e) ... and AT THE KERNEL PATH'S FORWARD POINT: its disparities and pose matrices (1e-6 / 3e-8 from the oracle's) replace
    # the oracle's values, the gradient path stays.  The loss is so ill-conditioned that this rounding of the FORWARD pass,
    # coherent over all pixels of a frame, is most of (d); what remains here is the arithmetic of the BACKWARD pass alone.
    point = {**{('disp', s): outs['disp', s].detach().cpu() for s in range(4)},
             **{('cam_T_cam', 0, f): outs['cam_T_cam', 0, f].detach().cpu() for f in (-1, 1)}}
    exact_pt = exact_run(point)
    fp32_pt = exact_run(point, double=False)
    # (f) ... and with the kernel path's fp32 PROJECTION MATRIX (K T)[:3] as part of the forward point: it is a rounded product
    # too (one ulp per entry moves every sample of a frame by ~1e-5 px, coherently), and the pose gradient is the most
    # sensitive quantity of the step to exactly that
    point_p = {**point, **{('P', f): ws.P[fi].detach().cpu() for fi, f in enumerate((-1, 1))}}
    exact_pp = exact_run(point_p)
    fp32_pp = exact_run(point_p, double=False)
    rows = [(name, rel_l2(hip[name], ref[name]), rel_l2(hip[name], forced[name]), float(ref[name].norm()),
             rel_l2(hip[name], forced3[name]), rel_l2(hip[name], exact[name]), rel_l2(forced3[name], exact[name]),
             rel_l2(hip[name], exact_pt[name]), rel_l2(fp32_pt[name], exact_pt[name]), rel_l2(hip[name], exact_pp[name]),
             rel_l2(fp32_pp[name], exact_pp[name])) for name in hip]
    return dict(flips=flips, npix=4 * B * H * W, gap=worst_gap, rows=rows, cell_flips=cell_flips, clip_flips=clip_flips, far_cells=far_cells, sign_flips=sign_flips,
                oracle_losses=ol, oracle_grads=ref, hip_grads=hip)


def report_attribution(tag, r) -> None:
    rows = r['rows']
    print(f"[{tag}] {r['flips']} of {r['npix']} selections differ (largest candidate gap {r['gap']:.2e}), {r['cell_flips']} "
          f"bilinear cells ({r['far_cells']} of them by more than one cell) and {r['clip_flips']} clip flags of {2 * r['npix']} samples; worst rel-L2 of the 36 gradient tensors "
          f"vs the oracle {max(x[1] for x in rows):.2e}, on the same selection {max(x[2] for x in rows):.2e}, on the same "
          f"selection + cells + clips {max(x[4] for x in rows):.2e}; against the float64 oracle on those decisions: kernels "
          f"{max(x[5] for x in rows):.2e}, torch fp32 {max(x[6] for x in rows):.2e}; the same at the kernel path's forward point with "
          f"its {r['sign_flips']} differing L1 signs imposed (backward arithmetic only): kernels {max(x[7] for x in rows):.2e}, torch fp32 {max(x[8] for x in rows):.2e}; "
          f"with the kernel path's rounded projection matrices part of that point: kernels {max(x[9] for x in rows):.2e}, torch fp32 {max(x[10] for x in rows):.2e}")
    shown = sorted(rows, key=lambda x: -x[1])[:4]
    shown += [x for x in sorted(rows, key=lambda x: -x[4])[:3] if x not in shown]
    shown += [x for x in sorted(rows, key=lambda x: -x[7])[:2] if x not in shown]
    for x in shown:
        print(f'    {x[0]:44s} free {x[1]:.2e}  same selection {x[2]:.2e}  + cells/clips {x[4]:.2e}   vs float64: kernels {x[5]:.2e}  '
              f'torch fp32 {x[6]:.2e}   at the same forward point: kernels {x[7]:.2e}  torch fp32 {x[8]:.2e}   + same P: kernels {x[9]:.2e}  torch fp32 {x[10]:.2e}')
