"""ResNet-18/50/101 definitions with torchvision-compatible state-dict keys.

torchvision is not installed in the build image and the reference's model loader
(pleas/datasets/model_loader.py:6-90) only wraps torchvision constructors, so the
benchmark workload (BASELINE.json configs) needs its own definitions.  Attribute
names and call order follow the public torchvision layout so that (a) real
torchvision checkpoints load with ``load_state_dict`` and (b) ``torch.fx`` gives
the node names a PermutationSpec refers to (``layer1_0_relu_1``, ``add_3`` ...).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Type

import torch
from torch import nn


def _conv(cin: int, cout: int, k: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None):
        super().__init__()
        self.conv1 = _conv(cin, planes, 3, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(planes, planes, 3)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None):
        super().__init__()
        self.conv1 = _conv(cin, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class ResNet(nn.Module):
    def __init__(self, block: Type[nn.Module], layers: Sequence[int], num_classes: int = 1000, width: int = 64):
        super().__init__()
        self._cin = width
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._stage(block, width, layers[0], 1)
        self.layer2 = self._stage(block, width * 2, layers[1], 2)
        self.layer3 = self._stage(block, width * 4, layers[2], 2)
        self.layer4 = self._stage(block, width * 8, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(width * 8 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _stage(self, block, planes: int, count: int, stride: int) -> nn.Sequential:
        downsample = None
        if stride != 1 or self._cin != planes * block.expansion:
            downsample = nn.Sequential(
                _conv(self._cin, planes * block.expansion, 1, stride),
                nn.BatchNorm2d(planes * block.expansion),
            )
        blocks: List[nn.Module] = [block(self._cin, planes, stride, downsample)]
        self._cin = planes * block.expansion
        blocks += [block(self._cin, planes) for _ in range(1, count)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


def resnet18(num_classes: int = 1000, width: int = 64) -> ResNet:
    return ResNet(BasicBlock, (2, 2, 2, 2), num_classes, width)


def resnet50(num_classes: int = 1000, width: int = 64) -> ResNet:
    return ResNet(Bottleneck, (3, 4, 6, 3), num_classes, width)


def resnet101(num_classes: int = 1000, width: int = 64) -> ResNet:
    return ResNet(Bottleneck, (3, 4, 23, 3), num_classes, width)


def tiny_resnet(block: str = "basic", layers: Sequence[int] = (1, 1, 1, 1), num_classes: int = 10, width: int = 8) -> ResNet:
    """Narrow/shallow variant used by tests and fixtures (same topology family)."""
    return ResNet(BasicBlock if block == "basic" else Bottleneck, tuple(layers), num_classes, width)


MODELS = {"resnet18": resnet18, "resnet50": resnet50, "resnet101": resnet101}


@torch.no_grad()
def calibrate_bn(model: nn.Module, batches, momentum: Optional[float] = None) -> nn.Module:
    """Run train-mode forwards so BN running stats match the data, then ``eval()``.

    Random-init deep ResNets in eval mode blow activations up by ~1e4 at layer4
    (SURVEY.md section 8(d)); the benchmark calibrates each source model on a few
    synthetic batches first.  ``momentum=None`` gives the cumulative average.
    """
    saved = {}
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            saved[m] = m.momentum
            m.reset_running_stats()
            m.momentum = momentum
    model.train()
    for x in batches:
        model(x)
    for m, mom in saved.items():
        m.momentum = mom
    return model.eval()
