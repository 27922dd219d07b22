"""Measurement tools only: CLSLAM_TOOL_LIB=<tag> binds cl-slam_amd/lib/variants/libclslam_hip_<tag>.so (tools/build_variant.py:
trace / probe builds of single kernels) instead of the production library.  Import BEFORE anything calls clslam_hip.get_lib()."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'cl-slam_amd'))
from clslam_hip import _lib  # noqa: E402

_tag = os.environ.get('CLSLAM_TOOL_LIB')
if _tag:
    _lib.LIB_PATH = ROOT / 'cl-slam_amd' / 'lib' / 'variants' / f'libclslam_hip_{_tag}.so'
    print(f'# tool library variant: {_lib.LIB_PATH.name}', file=sys.stderr)
