#!/usr/bin/env python
"""bench.py on a probe build of the kernel library (CLSLAM_TOOL_LIB=<tag>, tools/build_variant.py): A/B of a kernel variant inside
the whole step.  Measurement tool; the product and the driver's bench.py only ever load lib/libclslam_hip.so."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _variant  # noqa: F401,E402
import bench  # noqa: E402
bench.main()
