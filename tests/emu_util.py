"""Backends for the kernel-level tests.

``emu``: the kernel SOURCES compiled for the host against tests/emu (fibers emulate the GPU
         threads, MFMA/wave collectives follow the documented gfx950 lane layouts) -- validates
         indexing, math and the Python sequencing in a container without a GPU (-m "not gpu").
``hip``: the real gfx950 library on cuda:0 (-m gpu) -- the parity tests proper.
"""
import functools
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]

BACKENDS = [pytest.param('emu', id='emu'), pytest.param('hip', id='hip', marks=pytest.mark.gpu)]


@functools.lru_cache(maxsize=None)
def _emu_path():
    sys.path.insert(0, str(ROOT / 'tests' / 'emu'))
    import build_emu
    return build_emu.build()


def use_backend(name: str) -> torch.device:
    from clslam_hip import _lib
    if name == 'hip':
        _lib._LIB = None
        lib = _lib.get_lib()
        assert lib.is_device
        return torch.device('cuda:0')
    lib = _lib.install_library_for_tests(_emu_path())
    assert not lib.is_device
    return torch.device('cpu')
