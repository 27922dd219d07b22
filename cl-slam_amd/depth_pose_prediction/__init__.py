"""Drop-in replacement of the reference's ``depth_pose_prediction`` package for the online
adaptation / prediction hot path, executing on hand-written HIP kernels for MI355X (gfx950).
Same exports as the reference's depth_pose_prediction/__init__.py:1-3."""
import depth_pose_prediction.utils
from depth_pose_prediction.config import DepthPosePrediction as Config
from depth_pose_prediction.depth_pose_prediction import DepthPosePrediction

__all__ = ['Config', 'DepthPosePrediction', 'utils']
