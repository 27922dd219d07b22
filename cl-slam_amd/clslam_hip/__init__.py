"""clslam_hip: Python host side of the MI355X-native CL-SLAM depth/pose hot path.

``_lib``    ctypes binding of the C ABI in include/clslam_hip.h (libclslam_hip.so, gfx950)
``ops``     one thin wrapper per entry point (tensor pointers/shape plumbing only)
``engine``  the per-frame forward / backward / Adam schedule over a pre-planned HBM workspace
``synth``   deterministic synthetic weights and sample dicts (bench / tests)
"""
