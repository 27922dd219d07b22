"""clslam_copy_multi: many device-to-device copies in one launch (the hipGraph path's input staging and output
hand-out).  Byte-exact; any alignment; more items than one launch holds; empty items; refuses anything that is not
a plain copy."""
import pytest
import torch

from clslam_hip import ops
from clslam_hip._lib import ClslamError
from emu_util import BACKENDS, use_backend


@pytest.mark.parametrize('backend', BACKENDS)
def test_copy_multi_byte_exact(backend):
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(3)
    pairs, expect = [], []
    # 16-byte path, 4-byte path (odd float count), byte path (uint8 at odd offsets), float64, empty, > 24 items
    sizes = [4096, 1, 3, 17, 250000, 0, 64] + [5 + i for i in range(30)]
    for n in sizes:
        src = torch.randn(n, generator=g).to(dev)
        dst = torch.full((n,), float('nan'), device=dev)
        pairs.append((src, dst)); expect.append(src.clone())
    raw_s = torch.randint(0, 256, (1000,), dtype=torch.uint8, generator=g).to(dev)
    raw_d = torch.zeros(1000, dtype=torch.uint8, device=dev)
    pairs.append((raw_s[1:998], raw_d[3:1000])); expect.append(raw_s[1:998].clone())      # misaligned both ways
    d64 = torch.randn(7, dtype=torch.float64, generator=g).to(dev)
    o64 = torch.zeros(7, dtype=torch.float64, device=dev)
    pairs.append((d64, o64)); expect.append(d64.clone())
    ops.copy_multi(pairs)
    for (src, dst), e in zip(pairs, expect):
        assert torch.equal(dst.cpu(), e.cpu())
    assert int(raw_d[:3].sum()) == 0                                                       # nothing written before the slice
    ops.copy_multi([])                                                                     # no-op


@pytest.mark.parametrize('backend', BACKENDS)
def test_copy_multi_refuses_conversions(backend):
    dev = use_backend(backend)
    a = torch.zeros(8, device=dev)
    with pytest.raises(ClslamError):
        ops.copy_multi([(a, torch.zeros(8, dtype=torch.float64, device=dev))])
    with pytest.raises(ClslamError):
        ops.copy_multi([(a, torch.zeros(9, device=dev))])
    with pytest.raises(ClslamError):
        ops.copy_multi([(torch.zeros(4, 4, device=dev).t(), torch.zeros(4, 4, device=dev))])
