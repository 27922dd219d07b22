"""Steps 2..5 of the reference's shipped `adaptation_epochs: 5` (config/config_adapt.yaml:53, dpp.py:309-313) with RESOLVING POWER
(review r4 item 3 / weak #1).

tests/test_trajectory.py follows both implementations along their OWN five-step trajectories; Adam's first `lr * sign(g)` updates
make that chaotic (16-20 % of the updates flipped after two steps in ANY fp32 arithmetic), so its later steps can only hold an
order of magnitude.  Here every step is TEACHER-FORCED instead: the float64 oracle runs the five steps; before step s its complete
state -- trainable weights, Adam's exp_avg / exp_avg_sq / step count -- is rounded to float32 and installed in the float32 oracle
(three realisations: plain, and two whose weights and convolution outputs are moved by an ulp) AND in the HIP predictor; each runs
ONE step from there.  Nothing accumulates: the step's forward must agree with the float64 step to fp32 round-off (a wrong Adam
moment, a stale frozen feature or a mis-scaled bias correction at step s shows at once), and its update is compared element by
element in units of lr:

    forward  (disparity, both poses: relative L2; loss: relative)      <= 3e-6 / 3e-7 / 3e-6, and <= 3 x the oracle's worst
    update   fraction of elements whose new value is > lr/2 off        <= 2 x max over the oracle realisations + 1e-4
             mean |dw| / lr                                            <= 2 x max over the oracle realisations + 1e-3

with NO saturated / x10 branch.  Step 1 (fresh moments: the update IS lr * sign(g), every gradient entry inside the rounding noise
flips) is reported and held to its own, wider floor -- see DESIGN.md section 2 for why the HIP path flips ~3x more such entries
than torch's CPU kernels do there."""
import copy
import math

import pytest
import torch

from clslam_hip import synth
from clslam_hip.engine import TrainableLayout
from emu_util import BACKENDS, use_backend
from helpers import make_oracle, rel_l2
from predictor_util import make_predictor

STEPS = 5
LR = 1e-4
TRAINED = ('depth_decoder', 'pose_decoder')


def _snapshot(o):
    return {'models': {n: {k: v.detach().clone() for k, v in o.models[n].state_dict().items()} for n in TRAINED},
            'opt': copy.deepcopy(o.optimizer.state_dict())}


def _to_f32(snap):
    def cast(v):
        return v.float() if torch.is_tensor(v) and v.is_floating_point() else v
    opt = copy.deepcopy(snap['opt'])
    for st in opt['state'].values():
        for k in st:
            st[k] = cast(st[k])
    return {'models': {n: {k: cast(v) for k, v in sd.items()} for n, sd in snap['models'].items()}, 'opt': opt}


def _install(p, snap, oracle: bool):
    for n in TRAINED:
        p.models[n].load_state_dict(snap['models'][n])
    p.optimizer.load_state_dict(copy.deepcopy(snap['opt']))


def _dist(a, b):
    flipped = total = 0
    mean_lr = 0.0
    for name, wb in b['w'].items():
        d = (a['w'][name].double() - wb.double()).abs()
        flipped += int((d > 0.5 * LR).sum())
        total += d.numel()
        mean_lr += float(d.sum())
    return {'disp0': rel_l2(a['disp0'], b['disp0']), 'T-1': rel_l2(a['T-1'], b['T-1']), 'T+1': rel_l2(a['T+1'], b['T+1']),
            'loss': abs(a['loss'] - b['loss']) / abs(b['loss']), 'w_flipped': flipped / total, 'w_mean_lr': mean_lr / total / LR}


def _hip_step(p, batch, noise):
    p.set_tie_break_noise(noise)
    out, losses = p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
    eng = p.engine
    w = {name: TrainableLayout.to_reference(eng.w[off:off + math.prod(shape)], shape).cpu().clone()
         for name, off, shape in eng.layout.entries}
    return {'disp0': out['disp', 0].cpu().clone(), 'T-1': out['cam_T_cam', 0, -1].cpu().clone(),
            'T+1': out['cam_T_cam', 0, 1].cpu().clone(), 'loss': float(losses['loss']), 'w': w}


FWD = {'disp0': 3e-6, 'T-1': 3e-7, 'T+1': 3e-7, 'loss': 3e-6}      # (measured: 3-6e-7, 2-6e-8, 0-3e-7)


def _run(backend, H, W, B, seed, capsys, steps=STEPS, realisations=3):
    STEPS = steps                      # (shadows the module constant: the full-size minibatch runs fewer steps)
    use_backend(backend)
    batch = synth.make_batch(B, H, W, seed=seed)
    noises = [synth.make_noise(B, H, W, seed=seed + 20 + it) for it in range(STEPS)]
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    n64 = [{s: v.double() for s, v in n.items()} for n in noises]
    exact = make_oracle(H, W, B).to_double()
    snaps, recs = [], []
    for it in range(STEPS):
        snaps.append(_to_f32(_snapshot(exact)) if it else None)          # step 1 starts from the common closed-form weights
        recs.append(exact.trajectory(b64, n64[it:it + 1], 1)[0])

    def realisation(k):
        o = make_oracle(H, W, B)
        if k:
            gen = torch.Generator().manual_seed(2000 + k)

            def hook(_m, _inp, out):           # another summation order: every convolution output up to an ulp away
                r = torch.randint(0, 3, out.shape, generator=gen)
                d = out.detach()
                moved = torch.where(r == 0, torch.nextafter(d, torch.full_like(d, float('-inf'))),
                                    torch.where(r == 2, torch.nextafter(d, torch.full_like(d, float('inf'))), d))
                return out + (moved - d)
            for m in o.models.values():
                for sub in m.modules():
                    if isinstance(sub, torch.nn.Conv2d):
                        sub.register_forward_hook(hook)
        return o

    lines, bad = [], []
    for it in range(STEPS):
        dos = []
        for k in range(realisations):
            o = realisation(k)
            if snaps[it] is not None:
                _install(o, snaps[it], True)
            dos.append(_dist(o.trajectory(batch, noises[it:it + 1], 1)[0], recs[it]))
        p = make_predictor(H, W, B)
        if snaps[it] is not None:
            _install(p, snaps[it], False)
        dh = _dist(_hip_step(p, batch, noises[it]), recs[it])
        do = {k: max(d[k] for d in dos) for k in dh}
        lines.append(f'[{backend} {H}x{W} B={B}] step {it + 1} from the float64 state: ' + '  '.join(
            f'{k} {dh[k]:.1e} (oracle fp32: ' + ' / '.join(f'{d[k]:.1e}' for d in dos) + ')' for k in dh))
        for k, tol in FWD.items():
            if dh[k] > tol or dh[k] > 3 * do[k] + 0.1 * tol:
                bad.append((it + 1, k, dh[k], do[k]))
        if it == 0:
            # fresh moments: sign(g) updates, reported; bounded by the step-1 rule of tests/test_trajectory.py
            if dh['w_flipped'] > 2 * do['w_flipped'] + 2e-3:
                bad.append((1, 'w_flipped', dh['w_flipped'], do['w_flipped']))
        else:
            if dh['w_flipped'] > 2 * do['w_flipped'] + 1e-4:
                bad.append((it + 1, 'w_flipped', dh['w_flipped'], do['w_flipped']))
            if dh['w_mean_lr'] > 2 * do['w_mean_lr'] + 1e-3:
                bad.append((it + 1, 'w_mean_lr', dh['w_mean_lr'], do['w_mean_lr']))
    with capsys.disabled():
        print('\n' + '\n'.join(lines))
    assert not bad, bad


@pytest.mark.parametrize('backend', BACKENDS)
def test_every_step_from_the_float64_state(backend, capsys):
    _run(backend, 64, 128, 3 if backend == 'hip' else 2, 31, capsys)


@pytest.mark.gpu
def test_every_step_from_the_float64_state_at_full_size(capsys):
    _run('hip', 192, 640, 1, 33, capsys)


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_every_step_from_the_float64_state_at_the_benchmark_minibatch(capsys):
    """the HEADLINE minibatch (B = 5 at 192x640) held step by step from the float64 state, like B = 1 above --
    all five steps against the three fp32 realisations of the oracle (round 6, MI355X: forward 4-7e-7 / 3-6e-8 at every step, the
    warm-moment updates inside the oracle's own fp32 spread)."""
    _run('hip', 192, 640, 5, 35, capsys)
