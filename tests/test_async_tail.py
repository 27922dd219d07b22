"""Asynchronous tail (engine.async_tail, switched on by enable_data_parallel): gradient reduction, the gradient
all-reduce and Adam of step N run on their own stream while the caller's stream goes on with the frozen encoders of
step N+1.  It must be invisible: the same sequence of adapt() calls gives bitwise the same weights, moments and
outputs with it on or off, whatever the caller does right after adapt() returns."""
import pytest
import torch

from clslam_hip import synth
from emu_util import use_backend
from predictor_util import make_predictor

H, W, B = 64, 128, 3


def _run(async_tail: bool, steps: int = 6):
    p = make_predictor(H, W, B)
    p.engine.async_tail = async_tail
    batches = [synth.make_batch(B, H, W, seed=40 + i) for i in range(3)]
    p.set_tie_break_noise(synth.make_noise(B, H, W, seed=9))
    dev = p.device
    cur = {k: v.to(dev).clone() for k, v in batches[0].items()}
    outs = []
    for i in range(steps):
        out, losses = p.adapt(None, cur, steps=1)
        outs.append((out['disp', 0].clone(), losses['loss'].clone()))
        for k in cur:                       # the caller reuses its buffers at once
            cur[k].copy_(batches[(i + 1) % 3][k].to(dev))
    pred = p.predict({k: v.clone() for k, v in batches[1].items()})      # reads the weights: must wait for the tail
    p.engine.wait_training()
    sd = p.optimizer.state_dict()
    return (p.engine.w.clone(), p.engine.m.clone(), p.engine.v.clone(), outs, pred['depth', 0].clone(),
            sd['state'][62]['exp_avg'].clone())


@pytest.mark.gpu
def test_async_tail_is_bitwise_identical_to_serial():
    use_backend('hip')
    ref = _run(False)
    for _ in range(2):                      # twice: a race would not necessarily show on one run
        got = _run(True)
        for a, b in zip(ref[:3], got[:3]):
            assert torch.equal(a, b)
        for (d0, l0), (d1, l1) in zip(ref[3], got[3]):
            assert torch.equal(d0, d1) and torch.equal(l0, l1)
        assert torch.equal(ref[4], got[4]) and torch.equal(ref[5], got[5])
