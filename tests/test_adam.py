"""Fused Adam kernel vs torch.optim.Adam (the optimizer the reference constructs at dpp.py:203)."""
import pytest
import torch

from clslam_hip import ops
from emu_util import BACKENDS, use_backend
from helpers import rel_err


@pytest.mark.parametrize('backend', BACKENDS)
def test_adam_matches_torch(backend):
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(0)
    n = 4099
    p = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], 1e-4)
    w = p.clone().to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-6, 0, (1,), generator=g)))
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step(w, grad.to(dev), m, v, 1e-4, step)
        # <= 1 ulp of the parameter (|p| < 4) -- the update itself is lr-sized (1e-4)
        assert float((w.cpu() - ref.detach()).abs().max()) <= 4.8e-7
    st = opt.state[ref]
    assert rel_err(m.cpu(), st['exp_avg']) < 1e-6
    assert rel_err(v.cpu(), st['exp_avg_sq']) < 1e-6


@pytest.mark.parametrize('backend', BACKENDS)
def test_reduction_fused_with_adam_is_bitwise_the_two_launches(backend):
    """Single-GPU path: clslam_reduce_multi_adam (gradient partials -> gradient arena -> Adam update, one launch) against
    clslam_reduce_multi followed by clslam_adam_step: identical gradient, weights and moments, bit for bit, over two
    adaptation steps of the whole predictor; a NaN guard leaves weights and moments alone but still reduces."""
    dev = use_backend(backend)
    from clslam_hip import synth
    from predictor_util import make_predictor
    H, W, B = 64, 128, 2
    batch = synth.make_batch(B, H, W, seed=3)
    res = []
    for fuse in (True, False):
        p = make_predictor(H, W, B)
        p.engine.fuse_adam = fuse
        p.set_tie_break_noise(synth.make_noise(B, H, W, seed=5))
        for _ in range(2):
            _, losses = p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
        e = p.engine
        res.append((e.w.clone(), e.g.clone(), e.m.clone(), e.v.clone(), losses['loss'].clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    # the raw entry point with a NaN guard: gradients reduced, nothing updated
    n, splits = 1024, 3
    part = torch.randn(splits, n, generator=torch.Generator().manual_seed(1)).to(dev)
    g, w = torch.zeros(n, device=dev), torch.ones(n, device=dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    table = ops.make_reduce_table([(part, g, n, splits)], dev)
    ops.reduce_multi_adam(table, 1, g, w, m, v, 1e-4, 1, guard=torch.full((1,), float('nan'), device=dev))
    pd = part.cpu().double()          # the split partials are summed in double, in split order, and rounded once
    assert torch.equal(g.cpu(), ((pd[0] + pd[1]) + pd[2]).float())
    assert torch.equal(w.cpu(), torch.ones(n)) and not m.any() and not v.any()
    ops.reduce_multi_adam(table, 1, g, w, m, v, 1e-4, 1, guard=torch.zeros(1, device=dev))
    w2, m2, v2 = torch.ones(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    ops.adam_step(w2, g, m2, v2, 1e-4, 1)
    assert torch.equal(w, w2) and torch.equal(m, m2) and torch.equal(v, v2)
