"""Oracle networks: torch-CPU restatement of the reference's four networks.

State-dict key names are identical to the reference's so that one set of weights feeds the
reference (golden generation), this oracle and the HIP product.

* ``ResnetEncoder``  <- depth_pose_prediction/networks/resnet_encoder.py:79-125 (wrapper,
  input normalisation :117, 5-level feature list :118-125) and :13-44 (multi-image stem).
  The ResNet-18 body is torchvision==0.11.1 ``models.resnet.ResNet`` / ``BasicBlock``
  (requirements.txt:6), which is NOT vendored in the reference; it is restated from the
  published definition: conv3x3(bias=False)-BN-ReLU-conv3x3-BN-(+identity | 1x1 s2 conv-BN)
  -ReLU, BN eps 1e-5, stem conv7x7 s2 p3, maxpool 3x3 s2 p1, stages [2,2,2,2] with widths
  64/128/256/512 and stride 2 at the first block of stages 2-4.
* ``DepthDecoder``   <- depth_pose_prediction/networks/depth_decoder.py:14-71 and
  layers.py:9-48 (ConvBlock / Conv3x3: ReflectionPad2d(1) + 3x3 conv with bias + ELU).
* ``PoseDecoder``    <- depth_pose_prediction/networks/pose_decoder.py:11-54.
"""
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

NUM_CH_ENC = (64, 64, 128, 256, 512)
NUM_CH_DEC = (16, 32, 64, 128, 256)


class BasicBlock(nn.Module):
    """torchvision 0.11.1 models.resnet.BasicBlock (expansion 1), restated."""

    def __init__(self, inplanes: int, planes: int, stride: int = 1) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                nn.BatchNorm2d(planes),
            )

    def forward(self, x: Tensor) -> Tensor:
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class _ResNet18(nn.Module):
    """Parameter layout of torchvision ResNet-18 incl. the unused ImageNet head ``fc``
    (the reference's checkpoints carry ``resnet.fc.*``)."""

    def __init__(self, num_input_images: int = 1) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(3 * num_input_images, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)


class ResnetEncoder(nn.Module):
    """resnet_encoder.py:79-125."""

    def __init__(self, num_input_images: int = 1) -> None:
        super().__init__()
        self.num_ch_encoder = NUM_CH_ENC
        self.resnet = _ResNet18(num_input_images)

    def forward(self, x: Tensor) -> List[Tensor]:
        features = []
        x = (x - 0.45) / 0.225  # resnet_encoder.py:117
        x = self.resnet.conv1(x)
        x = self.resnet.bn1(x)
        features.append(self.resnet.relu(x))
        features.append(self.resnet.layer1(self.resnet.maxpool(features[-1])))
        features.append(self.resnet.layer2(features[-1]))
        features.append(self.resnet.layer3(features[-1]))
        features.append(self.resnet.layer4(features[-1]))
        return features


class Conv3x3(nn.Module):
    """layers.py:28-48 (reflection pad 1 + 3x3 conv with bias)."""

    def __init__(self, cin: int, cout: int) -> None:
        super().__init__()
        self.conv = nn.Conv2d(int(cin), int(cout), 3)

    def forward(self, x: Tensor) -> Tensor:
        return self.conv(F.pad(x, (1, 1, 1, 1), mode='reflect'))


class ConvBlock(nn.Module):
    """layers.py:9-25 (Conv3x3 + ELU)."""

    def __init__(self, cin: int, cout: int) -> None:
        super().__init__()
        self.conv = Conv3x3(cin, cout)

    def forward(self, x: Tensor) -> Tensor:
        return F.elu(self.conv(x))


class DepthDecoder(nn.Module):
    """depth_decoder.py:14-71."""

    def __init__(self, scales: Tuple[int, ...] = (0, 1, 2, 3)) -> None:
        super().__init__()
        self.scales = scales
        for i in range(4, -1, -1):
            cin = NUM_CH_ENC[-1] if i == 4 else NUM_CH_DEC[i + 1]
            setattr(self, f'upconv_{i}_0', ConvBlock(cin, NUM_CH_DEC[i]))
            cin = NUM_CH_DEC[i] + (NUM_CH_ENC[i - 1] if i > 0 else 0)
            setattr(self, f'upconv_{i}_1', ConvBlock(cin, NUM_CH_DEC[i]))
        for s in self.scales:
            setattr(self, f'dispconv_{s}', Conv3x3(NUM_CH_DEC[s], 1))

    def forward(self, feats: List[Tensor]) -> Dict[Tuple[str, int], Tensor]:
        out = {}
        x = feats[-1]
        for i in range(4, -1, -1):
            x = getattr(self, f'upconv_{i}_0')(x)
            if i > 0:  # depth_decoder.py:57-62 (nearest to the skip's size, then cat)
                x = F.interpolate(x, size=feats[i - 1].shape[2:], mode='nearest')
                x = torch.cat([x, feats[i - 1]], 1)
            else:  # :64
                x = F.interpolate(x, scale_factor=2, mode='nearest')
            x = getattr(self, f'upconv_{i}_1')(x)
            if i in self.scales:  # :67-69
                out[('disp', i)] = torch.sigmoid(getattr(self, f'dispconv_{i}')(x))
        return out


class PoseDecoder(nn.Module):
    """pose_decoder.py:11-54 with num_input_features=1, num_frames_to_predict_for=2
    (depth_pose_prediction.py:135-137)."""

    def __init__(self) -> None:
        super().__init__()
        self.squeeze = nn.Conv2d(512, 256, 1)
        self.pose_0 = nn.Conv2d(256, 256, 3, 1, 1)
        self.pose_1 = nn.Conv2d(256, 256, 3, 1, 1)
        self.pose_2 = nn.Conv2d(256, 12, 1)

    def forward(self, last_feature: Tensor) -> Tuple[Tensor, Tensor]:
        out = F.relu(self.squeeze(last_feature))
        out = F.relu(self.pose_0(out))
        out = F.relu(self.pose_1(out))
        out = self.pose_2(out)
        out = out.mean(3).mean(2)  # pose_decoder.py:49
        out = 0.01 * out.view(-1, 2, 1, 6)  # :50
        return out[..., :3], out[..., 3:]
