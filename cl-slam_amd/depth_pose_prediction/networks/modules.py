"""Parameter containers of the four networks with the reference's state-dict layout.

The reference builds these as torch modules and runs them through ATen/cuDNN
(networks/resnet_encoder.py, depth_decoder.py, pose_decoder.py, layers.py).  Here the modules
only HOLD the parameters, in the reference's names and OIHW shapes, so checkpoints written by
either side load in the other (dpp.py:669-749); all arithmetic runs in the HIP engine
(clslam_hip.engine), to which ``forward`` delegates.  ``nn.Conv2d`` / ``nn.BatchNorm2d`` are used
purely as parameter holders (same default initialisation as the reference); their own forward is
never called on the hot path.
"""
import weakref
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch import Tensor, nn

NUM_CH_ENC = np.array([64, 64, 128, 256, 512])
NUM_CH_DEC = np.array([16, 32, 64, 128, 256])


class _EngineBacked(nn.Module):
    """Mixin: modules delegate compute to the engine and make sure parameter reads see the
    engine's current (adapted) weights."""

    def _bind(self, engine, name: str) -> None:
        object.__setattr__(self, '_engine_ref', weakref.ref(engine))
        object.__setattr__(self, '_engine_name', name)

    def _engine(self):
        ref = getattr(self, '_engine_ref', None)
        eng = ref() if ref is not None else None
        if eng is None:
            raise RuntimeError('network is not bound to a clslam_hip engine (construct it through '
                               'DepthPosePrediction); there is no torch fallback for this path')
        return eng

    def _sync(self) -> None:
        ref = getattr(self, '_engine_ref', None)
        eng = ref() if ref is not None else None
        if eng is not None:
            eng.sync_modules()

    # every parameter read the reference performs goes through one of these (dpp.py:186,680-688,
    # 716-731,813-819)
    def state_dict(self, *args, **kwargs):
        self._sync()
        return super().state_dict(*args, **kwargs)

    def parameters(self, recurse: bool = True):
        self._sync()
        return super().parameters(recurse)

    def named_parameters(self, *args, **kwargs):
        self._sync()
        return super().named_parameters(*args, **kwargs)


class _BasicBlock(nn.Module):
    """torchvision BasicBlock parameter layout (conv1,bn1,conv2,bn2[,downsample.0/.1])."""

    def __init__(self, inplanes: int, planes: int, stride: int = 1) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.stride = stride
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        else:
            self.downsample = None


class _ResNet18(nn.Module):
    def __init__(self, num_input_images: int) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(3 * num_input_images, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = nn.Sequential(_BasicBlock(64, 64), _BasicBlock(64, 64))
        self.layer2 = nn.Sequential(_BasicBlock(64, 128, 2), _BasicBlock(128, 128))
        self.layer3 = nn.Sequential(_BasicBlock(128, 256, 2), _BasicBlock(256, 256))
        self.layer4 = nn.Sequential(_BasicBlock(256, 512, 2), _BasicBlock(512, 512))
        self.fc = nn.Linear(512, 1000)  # unused ImageNet head, present in the reference's checkpoints
        for m in self.modules():  # resnet_encoder.py:39-44
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class ResnetEncoder(_EngineBacked):
    """networks/resnet_encoder.py:79-125 (ResNet-18 only: the shipped configuration; 34/50 raise)."""

    def __init__(self, num_layers: int, pretrained: bool, num_input_images: int = 1) -> None:
        super().__init__()
        self.num_ch_encoder = NUM_CH_ENC.copy()
        if num_layers != 18:
            raise ValueError(f'Could not find a ResNet model with {num_layers} layers.'
                             if num_layers not in (18, 34) else
                             'The MI355X-native path implements ResNet-18 (the reference adaptation config).')
        if num_input_images < 1:
            raise ValueError(f'Invalid value ({num_input_images}) for num_input_images.')
        # `pretrained` would download ImageNet weights in the reference (resnet_encoder.py:71-75,107);
        # there is no network here and load_model() overwrites them anyway.
        self.num_input_images = num_input_images
        self.resnet = _ResNet18(num_input_images)

    def forward(self, x: Tensor) -> List[Tensor]:
        return self._engine().run_encoder(self._engine_name, x)


class _Conv3x3(nn.Module):
    def __init__(self, cin: int, cout: int) -> None:
        super().__init__()
        self.conv = nn.Conv2d(int(cin), int(cout), 3)


class _ConvBlock(nn.Module):
    def __init__(self, cin: int, cout: int) -> None:
        super().__init__()
        self.conv = _Conv3x3(cin, cout)


class DepthDecoder(_EngineBacked):
    """networks/depth_decoder.py:14-71."""

    def __init__(self, num_ch_encoder: np.ndarray, scales: Tuple[int, ...] = (0, 1, 2, 3), use_skips: bool = True) -> None:
        super().__init__()
        if not use_skips or tuple(scales) != (0, 1, 2, 3):
            raise ValueError('The MI355X-native path implements use_skips=True, scales=(0,1,2,3).')
        self.scales = scales
        self.use_skips = use_skips
        self.num_output_channels = 1
        self.num_ch_encoder = num_ch_encoder
        self.num_ch_decoder = NUM_CH_DEC.copy()
        for i in range(4, -1, -1):
            cin = self.num_ch_encoder[-1] if i == 4 else self.num_ch_decoder[i + 1]
            setattr(self, f'upconv_{i}_0', _ConvBlock(cin, self.num_ch_decoder[i]))
            cin = self.num_ch_decoder[i] + (self.num_ch_encoder[i - 1] if i > 0 else 0)
            setattr(self, f'upconv_{i}_1', _ConvBlock(cin, self.num_ch_decoder[i]))
        for s in self.scales:
            setattr(self, f'dispconv_{s}', _Conv3x3(self.num_ch_decoder[s], 1))

    def forward(self, input_features: List[Tensor]) -> Dict[Tuple[str, int], Tensor]:
        return self._engine().run_depth_decoder(input_features)


class PoseDecoder(_EngineBacked):
    """networks/pose_decoder.py:11-54 with num_input_features=1, num_frames_to_predict_for=2."""

    def __init__(self, num_ch_encoder: np.ndarray, num_input_features: int = 1, num_frames_to_predict_for=None) -> None:
        super().__init__()
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        if num_input_features != 1 or num_frames_to_predict_for != 2:
            raise ValueError('The MI355X-native path implements num_input_features=1, num_frames_to_predict_for=2.')
        self.num_ch_encoder = num_ch_encoder
        self.num_input_features = num_input_features
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.squeeze = nn.Conv2d(int(num_ch_encoder[-1]), 256, 1)
        self.pose_0 = nn.Conv2d(256, 256, 3, 1, 1)
        self.pose_1 = nn.Conv2d(256, 256, 3, 1, 1)
        self.pose_2 = nn.Conv2d(256, 12, 1)

    def forward(self, input_features) -> Tuple[Tensor, Tensor]:
        return self._engine().run_pose_decoder([f[-1] for f in input_features])
