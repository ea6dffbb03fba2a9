"""SE-ResNet trunk containers (reference: layers/SE_Resnet.py:59-211).

Same tree as ``Resnet.py`` plus an ``se`` gate on the first block of every stage
(the one that carries the projection shortcut, SE_Resnet.py:199-202).
"""
import torch.nn as nn

from .Resnet import Bottleneck as _PlainBottleneck
from .Resnet import ResNet as _PlainResNet
from .SE_module import SELayer


class Bottleneck(_PlainBottleneck):
    def __init__(self, inplanes, planes, stride=1, downsample=None, reduction=False, norm_layer=nn.BatchNorm2d, dcn=None):
        super().__init__(inplanes, planes, stride, downsample, norm_layer=norm_layer, dcn=dcn)
        if reduction:                       # keep the reference's registration order: se before downsample
            proj = self.downsample
            del self.downsample
            self.se = SELayer(planes * 4)
            self.downsample = proj
        self.reduc = reduction


class SEResnet(_PlainResNet):
    def __init__(self, architecture, norm_layer=nn.BatchNorm2d, dcn=None, stage_with_dcn=(False, False, False, False)):
        self.block = Bottleneck
        super().__init__(architecture, norm_layer=norm_layer, dcn=dcn, stage_with_dcn=stage_with_dcn)

    def make_layer(self, block, planes, blocks, stride=1, dcn=None):
        block = Bottleneck
        proj = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            proj = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                 self._norm_layer(planes * block.expansion, momentum=0.1))
        seq = [block(self.inplanes, planes, stride, proj, reduction=proj is not None, norm_layer=self._norm_layer)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, norm_layer=self._norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*seq)
