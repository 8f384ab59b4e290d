from .resnet import (BasicResNetBlock, Bottleneck, CifarResNet, ResNet, ResNet18, ResNet18Cifar, ResNet34, ResNet50, ResNet101,  # noqa: F401
                     ResNet152, ResNet50_3343)
