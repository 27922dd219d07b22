"""Oracle for the loop-closure feature encoder: torchvision==0.11.1 ``mobilenet_v3_small`` cut at
the 'flatten' node (reference loop_closure_detection/encoder.py:13-33), restated in torch.

PARITY UNPINNED: torchvision's source and the ImageNet weights the reference downloads are not
available to the build (requirements.txt:6) and the reference has no test or
golden vector at this boundary.  The architecture below is the published MobileNetV3-small
definition as torchvision 0.11 implements it (state-dict keys ``features.N...`` included so a real
checkpoint loads); the HIP path is tested against THIS restatement with closed-form weights.
Test infrastructure only.
"""
from functools import partial

import torch
import torch.nn.functional as F
from torch import Tensor, nn

# (in, kernel, expanded, out, use_se, activation, stride)
SETTINGS = [(16, 3, 16, 16, True, 'RE', 2), (16, 3, 72, 24, False, 'RE', 2), (24, 3, 88, 24, False, 'RE', 1),
            (24, 5, 96, 40, True, 'HS', 2), (40, 5, 240, 40, True, 'HS', 1), (40, 5, 240, 40, True, 'HS', 1),
            (40, 5, 120, 48, True, 'HS', 1), (48, 5, 144, 48, True, 'HS', 1), (48, 5, 288, 96, True, 'HS', 2),
            (96, 5, 576, 96, True, 'HS', 1), (96, 5, 576, 96, True, 'HS', 1)]


def make_divisible(v: float, divisor: int = 8) -> int:
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


BN = partial(nn.BatchNorm2d, eps=0.001, momentum=0.01)


class ConvBNAct(nn.Sequential):
    def __init__(self, cin, cout, k, stride=1, groups=1, act=None):
        layers = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), BN(cout)]
        if act == 'RE':
            layers.append(nn.ReLU(inplace=True))
        elif act == 'HS':
            layers.append(nn.Hardswish(inplace=True))
        super().__init__(*layers)


class SqueezeExcitation(nn.Module):
    def __init__(self, cin, squeeze):
        super().__init__()
        self.fc1 = nn.Conv2d(cin, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, cin, 1)

    def forward(self, x: Tensor) -> Tensor:
        s = F.adaptive_avg_pool2d(x, 1)
        s = F.hardsigmoid(self.fc2(F.relu(self.fc1(s))))
        return s * x


class InvertedResidual(nn.Module):
    def __init__(self, cin, k, exp, cout, use_se, act, stride):
        super().__init__()
        self.use_res = stride == 1 and cin == cout
        layers = []
        if exp != cin:
            layers.append(ConvBNAct(cin, exp, 1, act=act))
        layers.append(ConvBNAct(exp, exp, k, stride, groups=exp, act=act))
        if use_se:
            layers.append(SqueezeExcitation(exp, make_divisible(exp // 4, 8)))
        layers.append(ConvBNAct(exp, cout, 1, act=None))
        self.block = nn.Sequential(*layers)

    def forward(self, x: Tensor) -> Tensor:
        y = self.block(x)
        return x + y if self.use_res else y


class MobileNetV3SmallFeatures(nn.Module):
    """features + avgpool + flatten (576-d), with torchvision's parameter names."""

    def __init__(self) -> None:
        super().__init__()
        layers = [ConvBNAct(3, 16, 3, 2, act='HS')]
        layers += [InvertedResidual(*s) for s in SETTINGS]
        layers.append(ConvBNAct(96, 576, 1, act='HS'))
        self.features = nn.Sequential(*layers)

    def forward(self, x: Tensor) -> Tensor:
        return torch.flatten(F.adaptive_avg_pool2d(self.features(x), 1), 1)


def feature_encoder(model: MobileNetV3SmallFeatures, image: Tensor) -> Tensor:
    """encoder.py:28-33: Normalize(mean, std) then the feature extractor."""
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    model.eval()
    with torch.no_grad():
        return model((image - mean) / std)
