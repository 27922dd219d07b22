"""Detached training step (Engine.main_stream, DepthPosePrediction.adapt): forward, backward and the optimizer step of a
training adapt() are enqueued on the engine's own stream; the caller's stream is ordered behind the forward and the first
three launches of the backward only (the last reads of the caller's minibatch), so the reads slam/slam.py:181-188 does next
return while the rest of the backward + Adam still run.  It must be invisible: the same sequence of calls gives bitwise the
same weights, moments, outputs and losses with it on or off, whatever the caller does right after adapt() returns --
overwrite its minibatch in place, drop it and allocate over it, call from another stream, read the arenas."""
import pytest
import torch

from clslam_hip import synth
from emu_util import use_backend
from predictor_util import make_predictor

pytestmark = pytest.mark.gpu
H, W, B = 192, 640, 3


def _run(detached: bool, frames: int = 5, on_side_stream: bool = False, steps_per_call: int = 1):
    p = make_predictor(H, W, B)
    torch.cuda.synchronize()
    p.engine.detached_training = detached
    batches = [synth.make_batch(B, H, W, seed=40 + i) for i in range(3)]
    p.set_tie_break_noise({s: v.cuda() for s, v in synth.make_noise(B, H, W, seed=9).items()})
    dev = p.device
    stream = torch.cuda.Stream() if on_side_stream else torch.cuda.current_stream()
    outs = []
    with torch.cuda.stream(stream):
        cur = {k: v.to(dev).clone() for k, v in batches[0].items()}
        for i in range(frames):
            out, losses = p.adapt(None, cur, steps=steps_per_call)
            # what slam.py reads back, at once
            outs.append((out['cam_T_cam', 0, 1][0].cpu().clone(), losses['loss'].cpu().clone(), out['disp', 0].clone(),
                         out['rgb', -1, 2].clone()))
            if i % 2 == 0:
                for k in cur:                       # the caller reuses its buffers at once ...
                    cur[k].copy_(batches[(i + 1) % 3][k].to(dev))
            else:                                   # ... or drops them and allocates over the freed blocks
                shapes = {k: (v.shape, v.dtype) for k, v in cur.items()}
                del cur
                junk = [torch.full(sh, 7.0, dtype=dt, device=dev) for sh, dt in shapes.values()]
                cur = {k: v.to(dev).clone() for k, v in batches[(i + 1) % 3].items()}
                del junk
            if i == 2:
                g_mid = p.engine.g.clone()          # an outside read of the gradient arena: ordered behind the step
        pred = p.predict({k: v.clone() for k, v in batches[1].items()})      # reads the weights: must wait for the step
        T, _ = p.predict_pose(batches[2]['rgb', 0, 0][0], batches[2]['rgb', 1, 0][0])
        sd = p.optimizer.state_dict()
        res = (p.engine.w.clone(), p.engine.m.clone(), p.engine.v.clone(), g_mid, pred['depth', 0].clone(), torch.from_numpy(T),
               sd['state'][62]['exp_avg'].clone())
    torch.cuda.synchronize()
    return res, outs


def _same(ref, got):
    for a, b in zip(ref[0], got[0]):
        assert torch.equal(a.cpu(), b.cpu())
    assert len(ref[1]) == len(got[1])
    for x, y in zip(ref[1], got[1]):
        for a, b in zip(x, y):
            assert torch.equal(a.cpu(), b.cpu())


def test_detached_step_is_bitwise_the_step_on_the_callers_stream():
    use_backend('hip')
    ref = _run(False)
    for _ in range(2):                      # twice: a race would not necessarily show on one run
        _same(ref, _run(True))
    _same(ref, _run(True, on_side_stream=True))


def test_detached_multi_step_adapt_and_nan_abort():
    """adapt(steps=3) (frozen-feature reuse inside the call) detached == on the caller's stream; a NaN loss raises like
    dpp.py:1115-1118, leaves the weights untouched and the predictor usable."""
    use_backend('hip')
    _same(_run(False, frames=3, steps_per_call=3), _run(True, frames=3, steps_per_call=3))
    p = make_predictor(H, W, 1)
    assert p.engine.detached_ok()
    batch = {k: v.cuda() for k, v in synth.make_batch(1, H, W, seed=5).items()}
    p.adapt(None, dict(batch))
    w0 = p.engine.w.clone()
    bad = dict(batch)
    bad['rgb', 0, 0] = torch.full_like(batch['rgb', 0, 0], float('nan'))
    with pytest.raises(RuntimeError, match='NaN loss'):
        p.adapt(None, bad)
    assert torch.equal(p.engine.w, w0) and p.engine.adam_step_count == 1
    out, losses = p.adapt(None, dict(batch))
    assert torch.isfinite(losses['loss']).all() and p.engine.adam_step_count == 2


def test_callers_stream_does_not_wait_for_the_whole_step():
    """The point of it: when adapt() has returned, a read-back on the caller's stream completes while the engine's stream
    is still busy with the step (B = 5 at 192x640: ~1.2 ms of backward + Adam behind the released point)."""
    use_backend('hip')
    p = make_predictor(H, W, 5)
    batch = {k: v.cuda() for k, v in synth.make_batch(5, H, W, seed=2).items()}
    for _ in range(3):
        p.adapt(None, dict(batch))
    torch.cuda.synchronize()
    early = 0
    for _ in range(10):
        out, losses = p.adapt(None, dict(batch))
        out['cam_T_cam', 0, 1][0].cpu()             # waits for the caller's stream only
        early += int(not p.engine._train_done.query())
        torch.cuda.synchronize()
    assert early >= 8, early
