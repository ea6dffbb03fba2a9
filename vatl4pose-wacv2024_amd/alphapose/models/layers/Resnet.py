"""ResNet trunk: parameter containers with the reference's module tree.

Attribute names / state-dict keys follow alphapose/models/layers/Resnet.py
(:57-128 Bottleneck, :131-211 ResNet) so reference checkpoints load strictly.
The torch sub-modules only *hold* parameters and buffers; ``forward`` runs the
hand-written gfx950 kernels through ``alphapose.models.hip_engine``.
"""
import torch.nn as nn

_BLOCKS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=nn.BatchNorm2d, dcn=None):
        super().__init__()
        if dcn is not None:
            raise NotImplementedError("DCN is outside the hot path (no shipped config uses it; SURVEY.md §2.1 row 3)")
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(planes, momentum=0.1)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)   # stride on the 3x3
        self.bn2 = norm_layer(planes, momentum=0.1)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = norm_layer(planes * 4, momentum=0.1)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        from alphapose.models import hip_engine
        return hip_engine.run_module_nchw(self, x)


class ResNet(nn.Module):
    def __init__(self, architecture, norm_layer=nn.BatchNorm2d, dcn=None, stage_with_dcn=(False, False, False, False)):
        super().__init__()
        if architecture in ("resnet18", "resnet34"):
            raise NotImplementedError("BasicBlock trunks are unusable with the pose heads (2048-channel input is hard-wired)")
        assert architecture in _BLOCKS
        if dcn is not None and any(stage_with_dcn):
            raise NotImplementedError("DCN is outside the hot path")
        self._norm_layer = norm_layer
        self.architecture = architecture
        self.block = Bottleneck
        self.layers = list(_BLOCKS[architecture])
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64, eps=1e-5, momentum=0.1, affine=True)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self.make_layer(Bottleneck, 64, self.layers[0])
        self.layer2 = self.make_layer(Bottleneck, 128, self.layers[1], stride=2)
        self.layer3 = self.make_layer(Bottleneck, 256, self.layers[2], stride=2)
        self.layer4 = self.make_layer(Bottleneck, 512, self.layers[3], stride=2)

    def make_layer(self, block, planes, blocks, stride=1, dcn=None):
        proj = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            proj = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                 self._norm_layer(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, proj, norm_layer=self._norm_layer)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, norm_layer=self._norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def stages(self):
        return [self.layer1, self.layer2, self.layer3, self.layer4]

    def forward(self, x):
        from alphapose.models import hip_engine
        return hip_engine.run_module_nchw(self, x)
