"""``FeatureEncoder`` with the reference's interface (loop_closure_detection/encoder.py:7-33:
``FeatureEncoder(device)``, ``.num_features == 576``, ``__call__(image) -> (B,576)``) running the
MobileNetV3-small forward on the HIP kernels (clslam_hip.lcd).

The reference builds ``torchvision.models.mobilenet_v3_small(pretrained=True)``, i.e. it downloads
ImageNet weights.  Here the weights must already be on disk in torchvision's state-dict format:
``weights=`` (path or state dict), or $CLSLAM_MOBILENETV3_WEIGHTS, or torch hub's cache file
``mobilenet_v3_small-047dcff4.pth``.  There is no CPU fallback.
"""
import os
from pathlib import Path
from typing import Dict, Optional, Union

import torch
from torch import Tensor

from clslam_hip.lcd import NUM_FEATURES, MobileNetV3SmallHIP


class FeatureEncoder:
    def __init__(self, device: torch.device, weights: Optional[Union[str, Path, Dict[str, Tensor]]] = None) -> None:
        self.device = torch.device(device)
        self.num_features = NUM_FEATURES
        if weights is None:
            cands = [os.environ.get('CLSLAM_MOBILENETV3_WEIGHTS'),
                     Path(torch.hub.get_dir()) / 'checkpoints' / 'mobilenet_v3_small-047dcff4.pth']
            weights = next((Path(c) for c in cands if c and Path(c).exists()), None)
            if weights is None:
                raise FileNotFoundError('MobileNetV3-small ImageNet weights not found (no network here): pass weights=, set '
                                        'CLSLAM_MOBILENETV3_WEIGHTS or place mobilenet_v3_small-047dcff4.pth in the torch '
                                        'hub cache')
        if not isinstance(weights, dict):
            weights = torch.load(weights, map_location='cpu')
        self.model = MobileNetV3SmallHIP(weights, self.device)

    def __call__(self, image: Tensor) -> Tensor:
        if image.dim() == 3:
            image = image.unsqueeze(0)
        return self.model(image)
