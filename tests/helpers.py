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
