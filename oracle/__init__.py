"""CPU oracle for the CL-SLAM depth_pose_prediction hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cl-slam_amd/`` (the product) imports this
package.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / reported CPU baseline.

It is a plain torch-CPU fp32 restatement (autograd supplies the reference gradients) of
the reference algorithm, every function citing the reference file:line it follows
(paths relative to the upstream repository root).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference
(``/root/reference`` with third-party stubs) in the build
container, runs ``predict()`` / ``adapt()`` on closed-form weights and inputs and commits
the outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this oracle
against those vectors.  Caveat: torchvision==0.11.1's ResNet-18 BasicBlock
source is not vendored in the reference, so the block arithmetic is the published
torchvision definition restated here and in the golden generator's stub;
MobileNetV3-small (loop-closure encoder) is "parity unpinned" (no source, no weights).
"""
from .networks import DepthDecoder, PoseDecoder, ResnetEncoder  # noqa: F401
from .predictor import OraclePredictor  # noqa: F401
