"""Restatement of the torchvision-0.8.2 ``resnet50`` body (ResNet v1.5) that the
reference instantiates at ``COTR/models/backbone.py:104-106`` and truncates with
``IntermediateLayerGetter`` at ``backbone.py:71``.

torchvision is a third-party dependency pinned by the reference
(``environment.yml:92`` -> torchvision=0.8.2) and is not vendored under
``/root/reference``; this file restates its published architecture:

* stem: 7x7/2 conv (3->64, pad 3, no bias) -> norm -> ReLU -> 3x3/2 max-pool (pad 1)
* ``layerN``: Bottleneck blocks (1x1 -> 3x3 -> 1x1 x4), the stride sits on the
  3x3 conv ("v1.5"), block 0 of a stage has a 1x1 (strided) ``downsample`` + norm,
  residual add then ReLU
* child module names ``conv1, bn1, relu, maxpool, layer1..layer4`` /
  ``conv1, bn1, conv2, bn2, conv3, bn3, downsample.{0,1}`` so state-dict keys are
  the ones a torchvision checkpoint (and the reference's checkpoints) carry.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).
"""
from collections import OrderedDict

import torch
from torch import nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, has_downsample, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if has_downsample:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                norm_layer(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet50(nn.Module):
    """resnet50(replace_stride_with_dilation=[False]*3, norm_layer=...) minus avgpool/fc."""

    def __init__(self, norm_layer, replace_stride_with_dilation=None, **_unused):
        super().__init__()
        if replace_stride_with_dilation is not None:
            assert not any(replace_stride_with_dilation), 'dilation is not on the COTR path'
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = 64
        for i, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
            layers = []
            for b in range(blocks):
                layers.append(Bottleneck(inplanes, planes, stride if b == 0 else 1, b == 0, norm_layer))
                inplanes = planes * 4
            setattr(self, f'layer{i + 1}', nn.Sequential(*layers))
        # torchvision's default init for conv weights
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')


class IntermediateLayerGetter(nn.ModuleDict):
    """Keeps children up to the last requested one, returns {new_name: activation}."""

    def __init__(self, model, return_layers):
        wanted = dict(return_layers)
        layers = OrderedDict()
        remaining = dict(return_layers)
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = wanted

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out
