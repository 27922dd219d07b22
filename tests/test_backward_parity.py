"""End-to-end backward parity, attributed: the gradients of all 36 trainable tensors of ONE adapt
step against the oracle's autograd, as full tensors (relative L2), plus the explanation DESIGN.md section 2 gives for the
residual demonstrated instead of asserted:

  (a) the 4-way min selection (`sel`) of the kernel path and of the oracle are compared pixel by pixel: they differ in a
      handful of pixels, and every one of them is a genuine near-tie (the two candidates the paths picked differ by
      less than the fp32 noise of the photometric maps);
  (b) with the oracle forced to the kernel path's selection, the remaining difference of every gradient tensor drops to
      the kernels' own accuracy.

64x128, B=2 on the emulator and on the GPU; 192x640, B=1 on the GPU."""
import math

import pytest
import torch

from clslam_hip import synth
from clslam_hip.engine import TrainableLayout
from emu_util import BACKENDS, use_backend
from helpers import make_oracle
from predictor_util import make_predictor


def _oracle_grads(o, batch, noise, B):
    o.set_adapt()
    out, losses = o.process_batch(batch, noise, None)
    o.optimizer.zero_grad()
    losses['loss'].backward()
    grads = {}
    for model in ('depth_decoder', 'pose_decoder'):
        for k, prm in o.models[model].named_parameters():
            grads[f'{model}/{k}'] = prm.grad.detach().clone()
    return out, losses, grads


def _rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _run(backend, H, W, B, seed):
    dev = use_backend(backend)
    p = make_predictor(H, W, B)
    o = make_oracle(H, W, B)
    batch = synth.make_batch(B, H, W, seed=seed)
    noise = synth.make_noise(B, H, W, seed=seed + 10)
    p.set_tie_break_noise(noise)
    out, losses = p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
    eng = p.engine
    eng.wait_training()
    hip = {name: TrainableLayout.to_reference(eng.g[off:off + math.prod(shape)], shape).cpu().clone()
           for name, off, shape in eng.layout.entries}
    sel_hip = eng._ws[B].sel.cpu().clone()                       # (4, B, H, W) u8
    _, ol, ref = _oracle_grads(o, batch, noise, B)
    assert abs(float(losses['loss']) - float(ol['loss'])) <= 1e-4 * abs(float(ol['loss']))
    # (a) selection flips: few, and all of them near-ties
    flips = 0
    worst_gap = 0.0
    for s in range(4):
        so, sh = o.last_sel[s], sel_hip[s].long()
        diff = so != sh
        flips += int(diff.sum())
        if diff.any():
            comb = o.last_combined[s]
            a = torch.gather(comb, 1, so.unsqueeze(1)).squeeze(1)[diff]
            b = torch.gather(comb, 1, sh.unsqueeze(1)).squeeze(1)[diff]
            worst_gap = max(worst_gap, float((b - a).abs().max()))
    npix = 4 * B * H * W
    # (b) the oracle on the kernel path's selection
    o2 = make_oracle(H, W, B)
    o2.forced_sel = {s: sel_hip[s] for s in range(4)}
    _, _, forced = _oracle_grads(o2, batch, noise, B)
    rows = []
    for name in hip:
        rows.append((name, _rel_l2(hip[name], ref[name]), _rel_l2(hip[name], forced[name]), float(ref[name].norm())))
    return flips, npix, worst_gap, rows


@pytest.mark.parametrize('backend', BACKENDS)
def test_gradients_match_oracle_and_the_residual_is_selection_flips(backend, capsys):
    flips, npix, gap, rows = _run(backend, 64, 128, 2, seed=3)
    with capsys.disabled():
        print(f'[{backend}] 64x128 B=2: {flips} of {npix} selections differ (largest candidate gap {gap:.2e}); '
              f'worst rel-L2 vs oracle {max(r[1] for r in rows):.2e}, vs oracle on the same selection {max(r[2] for r in rows):.2e}')
    assert flips <= 2e-4 * npix and gap < 5e-6, (flips, gap)
    for name, e_free, e_forced, norm in rows:
        assert e_free < 5e-2, (name, e_free)                 # full tensors (not norms / slices); dominated by the flips:
        assert e_forced < 5e-4, (name, e_forced)             # ... this is what is left once the selection is the same
    if flips:   # measured on the emulator: 2 flipped pixels of 65536 -> 1.3e-2 on one tensor, 3.2e-4 with them matched
        assert max(r[2] for r in rows) < 0.2 * max(r[1] for r in rows)


@pytest.mark.gpu
def test_gradients_full_size_on_gpu(capsys):
    flips, npix, gap, rows = _run('hip', 192, 640, 1, seed=5)
    with capsys.disabled():
        print(f'[hip] 192x640 B=1: {flips} of {npix} selections differ (largest candidate gap {gap:.2e}); '
              f'worst rel-L2 vs oracle {max(r[1] for r in rows):.2e}, vs oracle on the same selection {max(r[2] for r in rows):.2e}')
        for r in sorted(rows, key=lambda r: -r[1])[:5]:
            print(f'    {r[0]:44s} free {r[1]:.2e}  same-selection {r[2]:.2e}')
    assert flips <= 2e-4 * npix and gap < 5e-6, (flips, gap)
    for name, e_free, e_forced, norm in rows:
        assert e_free < 3e-2 and e_forced < 3e-2, (name, e_free, e_forced)
