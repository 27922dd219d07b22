"""The C-ABI library builds for gfx950, loads, and exports every symbol include/clslam_hip.h declares
(no compute call without a GPU)."""
import ctypes
import re

import torch  # noqa: F401  (before the library: same HIP runtime)
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / 'include' / 'clslam_hip.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(clslam_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    sys.path.insert(0, str(ROOT / 'cl-slam_amd' / 'csrc'))
    import build as hip_build
    lib = hip_build.build(verbose=False)
    cdll = ctypes.CDLL(str(lib))
    declared = _declared()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(cdll, s)]
    assert not missing, missing
    assert cdll.clslam_is_device_build() == 1
    from clslam_hip import _lib
    assert sorted(_lib.exported_symbols()) == declared


def test_product_loader_refuses_non_device_build():
    """There is no CPU fallback: the product loader rejects the emulator build and a missing file."""
    import pytest
    from clslam_hip import _lib
    sys.path.insert(0, str(ROOT / 'tests' / 'emu'))
    import build_emu
    with pytest.raises(_lib.ClslamError):
        _lib.Library(build_emu.build(), require_device=True)
    with pytest.raises(_lib.ClslamError):
        _lib.Library(ROOT / 'nonexistent.so', require_device=True)


def test_no_product_import_of_oracle():
    for f in (ROOT / 'cl-slam_amd').rglob('*.py'):
        src = f.read_text()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_binding_checks_the_abi_version():
    """include/clslam_hip.h, the library and the ctypes binding agree on CLSLAM_ABI_VERSION (a descriptor grew a field in round 5,
    four entry points changed a pointer type in round 4): a stale library is refused at load time instead of being handed
    structures of another layout."""
    import pytest
    from clslam_hip import _lib
    text = (ROOT / 'include' / 'clslam_hip.h').read_text()
    declared = int(re.search(r'#define\s+CLSLAM_ABI_VERSION\s+(\d+)', text).group(1))
    assert declared == _lib.ABI_VERSION
    lib = _lib.Library(_lib.LIB_PATH, require_device=True)
    assert lib.cdll.clslam_version() == declared
    assert lib.cdll.clslam_last_error_string() == lib.cdll.clslam_last_error()
    import ctypes as C
    assert C.sizeof(_lib.ConvDesc) == 7 * 8 + 15 * 4 + 4 + 8 + 4 + 4 + 8 + 8 + 8 + 4 + 4      # ... + weight_wino + cu_limit (+ tail padding)
    old = _lib.ABI_VERSION
    try:
        _lib.ABI_VERSION = old + 1
        # ... and a stale library also lacks the entry points added since: the version is compared BEFORE they are resolved
        _lib._SIGNATURES['clslam_entry_point_of_a_later_abi'] = []
        with pytest.raises(_lib.ClslamError, match='ABI version'):
            _lib.Library(_lib.LIB_PATH, require_device=True)
    finally:
        _lib.ABI_VERSION = old
        _lib._SIGNATURES.pop('clslam_entry_point_of_a_later_abi', None)


def test_a_stale_armed_handoff_event_does_not_block_the_next_arm():
    """a launch that fails its argument checks between clslam_handoff_arm and clslam_handoff_wait leaves the event armed;
    the next arm used to fail forever ('an armed event has not been waited on').  Now it replaces the stale event.  (Pointer
    bookkeeping only -- nothing is launched, the fake handles are never dereferenced.)"""
    from clslam_hip import _lib
    lib = _lib.Library(_lib.LIB_PATH, require_device=True)
    lib.call('clslam_handoff_arm', ctypes.c_void_p(0x1000))
    lib.call('clslam_handoff_arm', ctypes.c_void_p(0x2000))      # raised ClslamError before
    import pytest
    with pytest.raises(_lib.ClslamError, match='null event'):
        lib.call('clslam_handoff_arm', None)
