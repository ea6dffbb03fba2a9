"""Squeeze-and-excitation gate container (reference: layers/SE_module.py:9-24).

``fc`` = Linear(c, c/r) - ReLU - Linear(c/r, c) - Sigmoid.  In the HIP plan the two
Linear layers run as 1x1 convolutions over a (B,1,1,C) tensor on the MFMA kernel and
the sigmoid gate is applied together with the residual add and ReLU.
"""
from torch import nn


class SELayer(nn.Module):
    def __init__(self, channel, reduction=1):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel), nn.Sigmoid())
