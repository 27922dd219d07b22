"""Same export list as the reference's depth_pose_prediction/networks/__init__.py:1-8."""
from depth_pose_prediction.networks.layers import SSIM, BackprojectDepth, Project3D
from depth_pose_prediction.networks.modules import DepthDecoder, PoseDecoder, ResnetEncoder

__all__ = ['SSIM', 'BackprojectDepth', 'DepthDecoder', 'PoseDecoder', 'Project3D', 'ResnetEncoder']
