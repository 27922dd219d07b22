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
