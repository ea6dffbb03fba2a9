"""HRNet behind the reference's builder surface (reference: alphapose/models/hrnet.py:24-456).

Parameter containers reproduce the reference's module tree (``conv1/bn1/conv2/bn2``,
``layer1``, ``transition{1,2,3}``, ``stage{2,3,4}[m].branches / .fuse_layers``,
``final_layer``: 1754 state-dict tensors for W32); the forward is a HIP plan in
``hip_engine``: every conv on the MFMA kernel with BN/ReLU/residual in its epilogue, the
nearest-upsample sums of a fusion row in one ``vatl_fuse_upsample_add`` launch.
"""
import torch.nn as nn

from .builder import SPPE

BN_MOMENTUM = 0.1


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


blocks_dict = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def _seq_conv_bn(cin, cout, k, stride, relu):
    mods = [nn.Conv2d(cin, cout, k, stride, k // 2, bias=False), nn.BatchNorm2d(cout)]
    if relu:
        mods.append(nn.ReLU(True))
    return nn.Sequential(*mods)


class HighResolutionModule(nn.Module):
    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels, fuse_method, multi_scale_output=True):
        super().__init__()
        for what, lst in (("NUM_BLOCKS", num_blocks), ("NUM_CHANNELS", num_channels), ("NUM_INCHANNELS", num_inchannels)):
            if num_branches != len(lst):
                raise ValueError(f"NUM_BRANCHES({num_branches}) <> {what}({len(lst)})")
        self.num_inchannels = num_inchannels
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.branches = nn.ModuleList([self._one_branch(i, blocks, num_blocks, num_channels) for i in range(num_branches)])
        self.fuse_layers = self._make_fuse_layers()
        self.relu = nn.ReLU(True)

    def _one_branch(self, i, block, num_blocks, num_channels):
        down = None
        if self.num_inchannels[i] != num_channels[i] * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.num_inchannels[i], num_channels[i] * block.expansion, 1, bias=False),
                                 nn.BatchNorm2d(num_channels[i] * block.expansion, momentum=BN_MOMENTUM))
        seq = [block(self.num_inchannels[i], num_channels[i], 1, down)]
        self.num_inchannels[i] = num_channels[i] * block.expansion
        seq += [block(self.num_inchannels[i], num_channels[i]) for _ in range(1, num_blocks[i])]
        return nn.Sequential(*seq)

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        c = self.num_inchannels
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(c[j], c[i], 1, 1, 0, bias=False), nn.BatchNorm2d(c[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                elif j == i:
                    row.append(None)
                else:
                    row.append(nn.Sequential(*[_seq_conv_bn(c[j], c[i] if k == i - j - 1 else c[j], 3, 2, relu=(k != i - j - 1))
                                               for k in range(i - j)]))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_inchannels(self):
        return self.num_inchannels


@SPPE.register_module
class PoseHighResolutionNet(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        self.inplanes = 64
        self._preset_cfg = cfg["PRESET"]
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(Bottleneck, 64, 4)
        pre = [256]
        for s in (2, 3, 4):
            sc = cfg[f"STAGE{s}"]
            setattr(self, f"stage{s}_cfg", sc)
            block = blocks_dict[sc["BLOCK"]]
            chans = [c * block.expansion for c in sc["NUM_CHANNELS"]]
            setattr(self, f"transition{s - 1}", self._make_transition_layer(pre, chans))
            stage, pre = self._make_stage(sc, chans, multi_scale_output=(s != 4))
            setattr(self, f"stage{s}", stage)
        k = cfg["FINAL_CONV_KERNEL"]
        self.final_layer = nn.Conv2d(pre[0], self._preset_cfg["NUM_JOINTS"], kernel_size=k, stride=1, padding=1 if k == 3 else 0)
        self.pretrained_layers = cfg["PRETRAINED_LAYERS"]

    def _make_transition_layer(self, pre, cur):
        out = []
        for i, c in enumerate(cur):
            if i < len(pre):
                out.append(_seq_conv_bn(pre[i], c, 3, 1, True) if c != pre[i] else None)
            else:
                out.append(nn.Sequential(*[_seq_conv_bn(pre[-1], c if j == i - len(pre) else pre[-1], 3, 2, True)
                                           for j in range(i + 1 - len(pre))]))
        return nn.ModuleList(out)

    def _make_layer(self, block, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion, momentum=BN_MOMENTUM))
        seq = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def _make_stage(self, layer_config, num_inchannels, multi_scale_output=True):
        n = layer_config["NUM_MODULES"]
        block = blocks_dict[layer_config["BLOCK"]]
        mods = []
        for i in range(n):
            mods.append(HighResolutionModule(layer_config["NUM_BRANCHES"], block, layer_config["NUM_BLOCKS"], num_inchannels,
                                             layer_config["NUM_CHANNELS"], layer_config["FUSE_METHOD"],
                                             multi_scale_output or i != n - 1))
            num_inchannels = mods[-1].get_num_inchannels()
        return nn.Sequential(*mods), num_inchannels

    def _initialize(self, pretrained=""):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        from . import hip_engine
        return hip_engine.run_module_nchw(self, x)
