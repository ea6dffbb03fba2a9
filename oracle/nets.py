"""Plain ``torch.nn`` (CPU, fp32) restatement of the reference graphs (TEST ORACLE).

These modules reproduce the *module tree* of the reference (same attribute
names, hence the same ``state_dict`` keys and shapes) so that a synthetic or
reference-produced state dict loads strictly, and their forward is the CPU
baseline timed by ``bench.py``.  They are never imported by the product.

SimplePoseRef     alphapose/models/simplepose.py:13-91 + layers/Resnet.py:57-211
WholeBodyAERef    active_learning/Whole_body_AE/AutoEncoder.py:5-39 with a
                  parametrised input width (38 | 42 | 51)
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

_DEPTHS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


class _Bneck(nn.Module):
    """1x1 -> 3x3 (carries the stride) -> 1x1(x4), residual add, ReLU.
    Resnet.py:57-128 (non-DCN branch)."""

    def __init__(self, cin: int, width: int, stride: int, project: bool):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, 4 * width, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * width)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, 4 * width, 1, stride, bias=False),
                                            nn.BatchNorm2d(4 * width))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        s = x if self.downsample is None else self.downsample(x)
        return F.relu(y + s)


class _Trunk(nn.Module):
    """7x7/2 stem, 3x3/2 max-pool, four bottleneck stages (Resnet.py:131-211)."""

    def __init__(self, depth: int):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for s, (n, width) in enumerate(zip(_DEPTHS[depth], (64, 128, 256, 512)), start=1):
            blocks = []
            for b in range(n):
                stride = 2 if (b == 0 and s > 1) else 1
                blocks.append(_Bneck(cin, width, stride, project=(b == 0)))
                cin = 4 * width
            setattr(self, f"layer{s}", nn.Sequential(*blocks))

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class SimplePoseRef(nn.Module):
    """ResNet trunk + 3 x (deconv 4x4/2 p1, BN, ReLU) + 1x1 head."""

    def __init__(self, num_layers: int = 50, deconv_filters=(256, 256, 256), num_joints: int = 17):
        super().__init__()
        self.preact = _Trunk(num_layers)
        mods, cin = [], 2048
        for c in deconv_filters:
            mods += [nn.ConvTranspose2d(cin, c, 4, 2, 1, bias=False), nn.BatchNorm2d(c), nn.ReLU(inplace=True)]
            cin = c
        self.deconv_layers = nn.Sequential(*mods)
        self.final_layer = nn.Conv2d(cin, num_joints, 1)

    def forward(self, x):
        return self.final_layer(self.deconv_layers(self.preact(x)))

    def get_embedding(self, x):
        return torch.flatten(F.adaptive_avg_pool2d(self.preact(x), 1), 1)


class WholeBodyAERef(nn.Module):
    """D -> 24 -> 12 -> 7 -> z -> 7 -> 12 -> 24 -> D, ReLU between, Sigmoid last."""

    def __init__(self, z_dim: int = 4, input_dim: int = 42):
        super().__init__()
        dims = (input_dim, 24, 12, 7, z_dim)
        enc, dec = [], []
        for i in range(4):
            enc.append(nn.Linear(dims[i], dims[i + 1]))
            if i < 3:
                enc.append(nn.ReLU(True))
        for i in range(4, 0, -1):
            dec.append(nn.Linear(dims[i], dims[i - 1]))
            dec.append(nn.ReLU(True) if i > 1 else nn.Sigmoid())
        self.encoder = nn.Sequential(*enc)
        self.decoder = nn.Sequential(*dec)

    def forward(self, x):
        return self.decoder(self.encoder(x))
