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


# ---------------------------------------------------------------------------
# FastPose: SE-ResNet trunk + PixelShuffle + 2 x DUC + 3x3 head
# (alphapose/models/fastpose.py:15-73, layers/SE_Resnet.py:59-211, SE_module.py:9-24, DUC.py:9-29)
# ---------------------------------------------------------------------------

class _SEGate(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(c, c), nn.ReLU(inplace=True), nn.Linear(c, c), nn.Sigmoid())

    def forward(self, x):
        g = self.fc(x.mean(dim=(2, 3)))
        return x * g[:, :, None, None]


class _SEBneck(_Bneck):
    """Bottleneck whose first block per stage (the one with a projection) gates conv3's output."""

    def __init__(self, cin, width, stride, project):
        super().__init__(cin, width, stride, project)
        if project:                                   # the reference registers `se` before `downsample`
            proj = self.downsample
            del self.downsample
            self.se = _SEGate(4 * width)
            self.downsample = proj

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        if self.downsample is not None:
            y = self.se(y)
        s = x if self.downsample is None else self.downsample(x)
        return F.relu(y + s)


class _SETrunk(_Trunk):
    def __init__(self, depth: int):
        nn.Module.__init__(self)
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for s, (n, width) in enumerate(zip(_DEPTHS[depth], (64, 128, 256, 512)), start=1):
            blocks = []
            for b in range(n):
                stride = 2 if (b == 0 and s > 1) else 1
                blocks.append(_SEBneck(cin, width, stride, project=(b == 0)))
                cin = 4 * width
            setattr(self, f"layer{s}", nn.Sequential(*blocks))


class _DUC(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.pixel_shuffle(F.relu(self.bn(self.conv(x))), 2)


class FastPoseRef(nn.Module):
    def __init__(self, num_layers: int = 50, num_joints: int = 17, conv_dim: int = 128):
        super().__init__()
        self.preact = _SETrunk(num_layers)
        self.suffle1 = nn.PixelShuffle(2)
        self.duc1 = _DUC(512, 1024)
        self.duc2 = _DUC(256, 1024 if conv_dim == 256 else 512)
        self.conv_out = nn.Conv2d(conv_dim, num_joints, 3, 1, 1)

    def forward(self, x):
        return self.conv_out(self.duc2(self.duc1(self.suffle1(self.preact(x)))))

    def get_embedding(self, x):
        return torch.flatten(F.adaptive_avg_pool2d(self.preact(x), 1), 1)


# ---------------------------------------------------------------------------
# HRNet (alphapose/models/hrnet.py:24-456)
# ---------------------------------------------------------------------------

class _Basic(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(y)) + x)


def _cbr(cin, cout, k, stride, relu):
    mods = [nn.Conv2d(cin, cout, k, stride, k // 2, bias=False), nn.BatchNorm2d(cout)]
    if relu:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


class _HRModule(nn.Module):
    """Parallel BasicBlock branches, then every output resolution sums all branches
    (1x1+BN+nearest-up for lower resolutions, strided 3x3+BN chains for higher), ReLU."""

    def __init__(self, chans, nblocks, multi_scale_output=True):
        super().__init__()
        nb = len(chans)
        self.branches = nn.ModuleList([nn.Sequential(*[_Basic(c) for _ in range(n)]) for c, n in zip(chans, nblocks)])
        rows = []
        for i in range(nb if multi_scale_output else 1):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(chans[j], chans[i], 1, bias=False), nn.BatchNorm2d(chans[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                elif j == i:
                    row.append(None)
                else:
                    row.append(nn.Sequential(*[_cbr(chans[j], chans[i] if k == i - j - 1 else chans[j], 3, 2, relu=(k != i - j - 1))
                                               for k in range(i - j)]))
            rows.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(rows)

    def forward(self, xs):
        xs = [b(x) for b, x in zip(self.branches, xs)]
        out = []
        for i, row in enumerate(self.fuse_layers):
            y = xs[0] if i == 0 else row[0](xs[0])
            for j in range(1, len(xs)):
                y = y + (xs[j] if i == j else row[j](xs[j]))
            out.append(F.relu(y))
        return out


class HRNetRef(nn.Module):
    """HRNet-W32 by default (configs/posetrack21/hrnetw32_posetrack21.yaml:37-57)."""

    def __init__(self, stages=((1, (32, 64)), (4, (32, 64, 128)), (3, (32, 64, 128, 256))), nblocks=4, num_joints=17, final_kernel=1):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64)
        self.layer1 = nn.Sequential(*[_Bneck(64 if b == 0 else 256, 64, 1, project=(b == 0)) for b in range(4)])
        pre = [256]
        for s, (nmod, chans) in enumerate(stages, start=2):
            trans = []
            for i, c in enumerate(chans):
                if i < len(pre):
                    trans.append(_cbr(pre[i], c, 3, 1, True) if c != pre[i] else None)
                else:
                    trans.append(nn.Sequential(*[_cbr(pre[-1], c if j == i - len(pre) else pre[-1], 3, 2, True)
                                                 for j in range(i + 1 - len(pre))]))
            setattr(self, f"transition{s - 1}", nn.ModuleList(trans))
            last_stage = s == len(stages) + 1
            setattr(self, f"stage{s}", nn.Sequential(*[_HRModule(chans, [nblocks] * len(chans),
                                                                 multi_scale_output=not (last_stage and m == nmod - 1))
                                                       for m in range(nmod)]))
            pre = list(chans)
        self.final_layer = nn.Conv2d(pre[0], num_joints, final_kernel, 1, 1 if final_kernel == 3 else 0)
        self._nstage = len(stages)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.layer1(x)
        ys = [x]
        for s in range(2, self._nstage + 2):
            trans = getattr(self, f"transition{s - 1}")
            xs = []
            for i, t in enumerate(trans):
                if t is None:
                    xs.append(ys[i])
                else:
                    xs.append(t(ys[-1]))
            for mod in getattr(self, f"stage{s}"):
                xs = mod(xs)
            ys = xs
        return self.final_layer(ys[0])


# ---------------------------------------------------------------------------
# L1JointRegression (alphapose/models/criterion.py:13-94, transforms.py:645-702)
# ---------------------------------------------------------------------------

class _IntegralCoordinate(torch.autograd.Function):
    """criterion.py:13-43: forward multiplies by the index ramp, backward passes +-AMPLITUDE instead of the ramp."""

    @staticmethod
    def forward(ctx, inp):
        w = torch.arange(inp.shape[-1], dtype=inp.dtype)
        out = inp.mul(w)
        ctx.n = inp.shape[-1]
        ctx.save_for_backward(w, out)
        return out

    @staticmethod
    def backward(ctx, g):
        w, out = ctx.saved_tensors
        coord = out.sum(dim=2, keepdim=True)
        w = w[None, None, :].repeat(coord.shape[0], coord.shape[1], 1)
        mask = torch.ones_like(w)
        mask[w < coord] = -1
        mask[coord.repeat(1, 1, w.shape[-1]) > ctx.n] = 1
        return g.mul(mask * 2)


def l1_joint_regression(preds, gt_joints, gt_vis, norm_type="softmax", size_average=True):
    """L1JointRegression.forward (criterion.py:60-80) on (B,J,H,W) heat-maps; returns (loss, pred_jts)."""
    b, j, h, w = preds.shape
    p = preds.reshape(b, j, -1)
    if norm_type == "softmax":
        p = F.softmax(p, 2)
    elif norm_type == "sigmoid":
        p = p.sigmoid()
    else:
        p = p / p.sum(dim=2, keepdim=True)
    p = (p / p.sum(dim=2, keepdim=True)).reshape(b, j, 1, h, w)
    hm_x, hm_y = p.sum((2, 3)), p.sum((2, 4))
    cx = _IntegralCoordinate.apply(hm_x).sum(dim=2, keepdim=True) / float(w) - 0.5
    cy = _IntegralCoordinate.apply(hm_y).sum(dim=2, keepdim=True) / float(h) - 0.5
    jts = torch.cat((cx, cy), dim=2).reshape(b, j * 2)
    out = (torch.abs(jts - gt_joints) * gt_vis).sum()
    return (out / b if size_average else out), jts
