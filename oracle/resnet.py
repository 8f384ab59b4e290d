"""TEST INFRASTRUCTURE ONLY (the parity oracle) - never imported by the product package.

CPU restatement in plain PyTorch fp32 of the reference's ResNet family
(/root/reference/src/super_gradients/training/models/classification_models/resnet.py): BasicResNetBlock :26-50,
Bottleneck :53-84, CifarResNet :87-137, ResNet :140-210 - same module tree, so the state_dict keys equal the reference's
and `load_state_dict(reference.state_dict())` works both ways; pinned against the real reference by
tests/test_oracle_vs_reference.py (live, through oracle/ref_shim.py) and tests/golden/resnet*.pt (oracle/make_golden.py).
"""
import torch.nn.functional as F
from torch import nn


class BasicBlock(nn.Module):
    def __init__(self, cin, planes, stride, expansion=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or cin != expansion * planes:
            self.shortcut = nn.Sequential(nn.Conv2d(cin, expansion * planes, 1, stride, bias=False), nn.BatchNorm2d(expansion * planes))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + self.shortcut(x))


class BottleneckBlock(nn.Module):
    def __init__(self, cin, planes, stride, expansion=4):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, expansion * planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(expansion * planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or cin != expansion * planes:
            self.shortcut = nn.Sequential(nn.Conv2d(cin, expansion * planes, 1, stride, bias=False), nn.BatchNorm2d(expansion * planes))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + self.shortcut(x))


class ResNetOracle(nn.Module):
    """cifar=True: 3x3 s1 stem, no max-pool (CifarResNet); else 7x7 s2 stem + 3x3 s2 max-pool (ResNet)."""

    def __init__(self, block, layers, num_classes, expansion, cifar=False, in_channels=3):
        super().__init__()
        self.cifar = cifar
        self.conv1 = nn.Conv2d(in_channels, 64, 3, 1, 1, bias=False) if cifar else nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        if not cifar:
            self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2))):
            blocks = []
            for s in [stride] + [1] * (n - 1):
                blocks.append(block(cin, planes, s, expansion))
                cin = planes * expansion
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        if cifar:  # registration order of the reference: CifarResNet defines avgpool before linear (no parameters either way)
            self.avgpool = nn.AdaptiveAvgPool2d(1)
            self.linear = nn.Linear(512 * expansion, num_classes)
        else:
            self.linear = nn.Linear(512 * expansion, num_classes)
            self.avgpool = nn.AdaptiveAvgPool2d(1)

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        if not self.cifar:
            out = self.maxpool(out)
        out = self.layer4(self.layer3(self.layer2(self.layer1(out))))
        return self.linear(self.avgpool(out).flatten(1))


def build(name: str, num_classes: int):
    table = {"resnet18": (BasicBlock, [2, 2, 2, 2], 1, False), "resnet34": (BasicBlock, [3, 4, 6, 3], 1, False),
             "resnet50": (BottleneckBlock, [3, 4, 6, 3], 4, False), "resnet18_cifar": (BasicBlock, [2, 2, 2, 2], 1, True)}
    block, layers, exp, cifar = table[name]
    return ResNetOracle(block, layers, num_classes, exp, cifar)
