"""Inference plans: torch module tree -> sequence of libvatl_hip.so launches.

A plan is built once per (module, parameter version, device): conv weights are
re-packed K-contiguous ([Cout][R][S][Cin]), eval-mode BatchNorm is folded into a
per-channel scale/bias consumed by the conv epilogue (with residual add and ReLU),
ConvTranspose2d(4,2,1) becomes four sub-pixel 2x2 filters.  Activations stay NHWC
between launches; only the network input (NCHW crops) and the heat-maps (NCHW,
what the scorers and the reference's callers expect) are converted.

FastPose (fastpose.py:52-59) and HRNet (hrnet.py:421-456) plans reuse the same conv
launches plus vatl_pixelshuffle2_fwd / vatl_se_scale_add_relu / vatl_fuse_upsample_add.

Dataflow of SimplePose (reference: simplepose.py:82-86, Resnet.py:171-177):
  NCHW crops -> NHWC(4ch) -> stem 7x7/2 (+BN+ReLU) -> maxpool 3x3/2
  -> 16 bottlenecks {1x1, 3x3(stride), 1x1 (+proj) + add + ReLU}
  -> 3 x deconv4x4/2 (+BN+ReLU) -> 1x1 head (+bias) written NCHW.

There is no fallback: without the library or on CPU tensors this raises.
"""
from __future__ import annotations

import time
import weakref

import torch
import torch.nn as nn

import vatl_hip as vh

# items per launch sequence; bounds workspace (stem output = 3.1 MB/crop) and keeps
# every tensor below the kernels' 2^30-element guard (32-bit buffer byte offsets);
# large chunks keep every layer's tile grid a multiple of the 512 resident blocks
MAX_CHUNK = 1024
# (Running the stem / layer-1 stage in Infinity-Cache-sized sub-batches was measured neutral-to-negative on MI355X in round 1,
# profiles/r01_notes.md, and is gone.)

# ROUTE CONSTANTS.  None of the switches below is read from the environment: the product takes exactly one route per layer
# geometry.  Tools and tests that A/B a route set the module attribute (before the plan of a model is built) — INTEGRATION.md
# "Process-global state and switches" lists which of them change bits.


# 3x3 / stride 1 / pad 1 layers run as Winograd F(2x2, 3x3) (csrc/conv_winograd.hip: 2.25x fewer multiplies, fp32).  The choice depends
# on the layer's geometry only — never on the batch — so a crop's heat-map bits do not depend on how it was batched.  The one exception is the
# small-batch module call (`model(x)` with <= 16 crops, run_module_nchw below): since round 3 it runs with split-K BY DEFAULT (before that
# split-K was opt-in) and stays on the implicit GEMM for every batch size it serves (vh.latency_mode()), so
# `model(x)` gives one set of bits for <= 16 crops and the stream route's bits above; both hold the 1e-4 / arg-max contract
# (SPLITK_AUTO_MAX = 0 makes the module call take the stream route at every size).  WINOGRAD = False = the implicit GEMM everywhere (other bits).
WINOGRAD = True
# 1x1 layers with K = 128 and N a multiple of 128 (Bottleneck.conv3 of stage 2; conv3 + projection of stage 1's first block) through the row-streaming GEMM
# (csrc/conv1x1_rows.hip: filter slice in registers, 32-pixel tiles; bit-identical to the tiled kernels, which ROWS_GEMM = False selects).
ROWS_GEMM = True
# ... and, since round 6, K = 256 with N >= 512 (Bottleneck.conv3 of ResNet stage 3: 256 -> 1024 + skip; `conv1x1_rows256_kernel`, bit-identical as well).
ROWS_GEMM_K = (128, 256)
# 32 -> 32 channel 3x3 layers (HRNet's highest-resolution branch) through the wave-private Winograd kernel (csrc/winograd_c32.hip: a wave owns 16 tiles with all 16
# transform positions, no cross-wave exchange).  False = the general Winograd kernel (same values to fp32 rounding, not the same bits).
WINO_C32 = True
# 3x3 / stride 1 / pad 1 layers with >= 64 input channels, a multiple of 64 output channels and a grid of whole 4x4 tiles (ResNet stages 1 - 3 at 256x192, HRNet's 64- and
# 128-channel branches, the DUC convs) as Winograd F(4x4,3x3) (csrc/winograd_f4.hip: 0.5625x the multiplies of F(2x2); measured 1.17 - 1.47x at 1024 crops, tools/f4_bench.py).
# Geometry-only choice like WINOGRAD: a crop's bits do not depend on its batch.  False = the F(2x2) route (same values to fp32 rounding, not the same bits).
WINO_F4 = True


class _Conv:
    __slots__ = ("w", "u", "u32", "u4", "wsrc", "wino", "c32", "f4", "scale", "bias", "cout", "r", "s", "stride", "pad")

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d | None):
        assert conv.groups == 1 and conv.dilation == (1, 1)
        assert conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
        self.w = vh.pack_conv_weight(conv.weight.detach())
        self.cout, _, self.r, self.s = conv.weight.shape
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        # Winograd filters are packed on the FIRST call that takes their route (the grid — whole 4x4 tiles or not — is only known then): a layer
        # that never runs F(4x4) (ResNet stage 4 at 8x6; stages 3 - 4 of R152 at 384x288) does not keep 36 floats per channel pair for it, a layer
        # that always does keeps no F(2x2) filter (113 MB per SimplePose-R50 plan, 450 MB for R152, per replica; and the pack time after every
        # fine-tune).  `wsrc` is the module's own weight tensor (no copy); the plan is rebuilt — or the parameter guard raises — when it changes.
        self.u = self.u32 = self.u4 = None
        self.wino = WINOGRAD and (self.r, self.s, self.stride, self.pad) == (3, 3, 1, 1) and conv.in_channels % 16 == 0 and self.cout % 4 == 0
        self.c32 = self.wino and WINO_C32 and conv.in_channels == 32 and self.cout == 32
        self.f4 = self.wino and WINO_F4 and conv.in_channels >= 64 and conv.in_channels % 16 == 0 and self.cout % 64 == 0
        self.wsrc = conv.weight.detach() if self.wino else None
        cb = conv.bias.detach() if conv.bias is not None else None
        if bn is not None:
            self.scale, self.bias = vh.bn_fold(_d(bn.weight), _d(bn.bias), bn.running_mean, bn.running_var, bn.eps, cb)
        elif cb is not None:
            self.scale, self.bias = vh.bn_fold(None, None, None, None, 0.0, cb, channels=self.cout)
        else:
            self.scale = self.bias = None

    def __call__(self, x, relu, residual=None, out_nchw=False, out=None):
        if self.wino and not out_nchw and not vh.latency_mode():
            if self.c32 and vh.conv3x3_winograd_c32_supported(x.shape[0], x.shape[1], x.shape[2], 32, 32):
                if self.u32 is None:
                    self.u32 = vh.pack_winograd_c32_weight(self.wsrc)
                return vh.conv3x3_winograd_c32_fwd(x, self.u32, self.scale, self.bias, relu, residual=residual, out=out)
            if self.f4 and vh.conv3x3_winograd_f4_supported(x.shape[0], x.shape[1], x.shape[2], x.shape[3], self.cout):
                if self.u4 is None:
                    self.u4 = vh.pack_winograd_f4_weight(self.wsrc)
                return vh.conv3x3_winograd_f4_fwd(x, self.u4, self.scale, self.bias, self.cout, relu, residual=residual, out=out)
            if self.u is None:
                self.u = vh.pack_winograd_weight(self.wsrc)
            return vh.conv3x3_winograd_fwd(x, self.u, self.scale, self.bias, self.cout, relu, residual=residual, out=out)
        if (ROWS_GEMM and self.r == 1 and self.stride == 1 and not out_nchw and not vh.latency_mode() and x.shape[-1] in ROWS_GEMM_K
                and vh.conv1x1_rows_supported(x.shape[-1], 0, self.cout, x.shape[0] * x.shape[1] * x.shape[2])):
            return vh.conv1x1_rows_fwd(x, self.w, self.scale, self.bias, self.cout, relu, residual=residual, out=out)     # K = 128 / 256, wide N: row-streaming GEMM
        return vh.conv2d_fwd(x, self.w, self.scale, self.bias, self.cout, self.r, self.s, self.stride, self.pad, relu,
                             residual=residual, out_nchw=out_nchw, out=out)


class _Deconv:
    __slots__ = ("w", "u", "scale", "bias", "cout")

    def __init__(self, dc: nn.ConvTranspose2d, bn: nn.BatchNorm2d):
        assert dc.kernel_size == (4, 4) and dc.stride == (2, 2) and dc.padding == (1, 1) and dc.bias is None
        self.cout = dc.weight.shape[1]
        self.u = None
        if WINOGRAD and dc.weight.shape[0] % 16 == 0 and self.cout % 4 == 0:     # four 2x2 phase convolutions as Winograd F(3x3, 2x2)
            self.u = vh.pack_winograd_deconv_weight(dc.weight.detach())
        self.w = vh.pack_deconv_weight(dc.weight.detach())                      # the implicit GEMM serves the small-batch module calls
        self.scale, self.bias = vh.bn_fold(_d(bn.weight), _d(bn.bias), bn.running_mean, bn.running_var, bn.eps)

    def __call__(self, x, relu=True):
        if self.u is not None and not vh.latency_mode():
            return vh.deconv4x4s2_winograd_fwd(x, self.u, self.scale, self.bias, self.cout, relu)
        return vh.deconv4x4s2_fwd(x, self.w, self.scale, self.bias, self.cout, relu)


def _d(p):
    return None if p is None else p.detach()


# Fuse conv3 with the projection shortcut of a stage's first block into one dual-source GEMM (vatl_conv1x1_dual_fwd):
# the projection output (3.2 GB per 1024 crops in layer 1) is never written or re-read.  False = two launches (other summation order).
FUSE_PROJ = True


class _BottleneckPlan:
    def __init__(self, blk):
        self.c1 = _Conv(blk.conv1, blk.bn1)
        self.c2 = _Conv(blk.conv2, blk.bn2)
        self.c3 = _Conv(blk.conv3, blk.bn3)
        self.proj = _Conv(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
        self.dual = None
        if (FUSE_PROJ and self.proj is not None and type(self) is _BottleneckPlan and self.proj.r == 1 and self.c3.cout >= 128
                and blk.conv3.in_channels % 32 == 0 and blk.downsample[0].in_channels % 32 == 0):
            w, b = vh.pack_conv1x1_dual_weight(blk.conv3.weight.detach(), self.c3.scale, self.c3.bias, blk.downsample[0].weight.detach(),
                                               self.proj.scale, self.proj.bias)
            self.dual = (w, b, self.proj.stride)

    def __call__(self, x, out=None, y1=None):
        y = y1 if y1 is not None else self.c1(x, relu=True)         # y1: conv1 + bn1 + relu already computed by the block before (chained launch)
        y = self.c2(y, relu=True)
        return self.tail(y, x, out=out)

    def tail(self, y, x, out=None):
        if self.dual is not None:                                   # relu(bn3(conv3(y)) + bn_p(conv_p(x))) in one GEMM over K = C1 + C2
            if (ROWS_GEMM and self.dual[2] == 1 and y.shape[-1] == 64 and x.shape[-1] == 64 and not vh.latency_mode()
                    and vh.conv1x1_rows_supported(64, 64, self.c3.cout, y.shape[0] * y.shape[1] * y.shape[2])):
                return vh.conv1x1_rows_fwd(y, self.dual[0], None, self.dual[1], self.c3.cout, True, x2=x, out=out)
            return vh.conv1x1_dual_fwd(y, x, self.dual[0], self.dual[1], self.c3.cout, self.dual[2], True, out=out)
        skip = x if self.proj is None else self.proj(x, relu=False)
        return self.c3(y, relu=True, residual=skip, out=out)        # relu(bn3(conv3) + skip)


# conv3 + bn3 + skip + relu of an identity-shortcut bottleneck chained with the NEXT block's conv1 + bn1 + relu in one launch (csrc/bottleneck_chain.hip, the
# 64 -> 256 -> 64 shapes of ResNet stage 1 / HRNet layer1): the 256-channel tensor is written once and not re-read.  Not in the small-batch module calls.
# False = the separate launches (conv3's output bit-identical; the chained conv1 sums K in four pieces).
FUSE_CHAIN = True


def _run_blocks(blocks, x, out=None):
    """Walk consecutive residual blocks; `out` receives the last block's output when that block can write into it."""
    y1 = None
    for k, b in enumerate(blocks):
        last = k == len(blocks) - 1
        o = out if last else None
        if type(b) is not _BottleneckPlan:
            x, y1 = b(x), None
            continue
        m = x.shape[0] * x.shape[1] * x.shape[2]
        chain = (FUSE_CHAIN and b.proj is None and not vh.latency_mode() and b.c3.scale is not None
                 and vh.bottleneck_chain_supported(b.c3.w.shape[-1], b.c3.cout, 0, m))
        if not chain:
            x, y1 = b(x, out=o, y1=y1), None
            continue
        nxt = None if last else blocks[k + 1]
        link = (type(nxt) is _BottleneckPlan and (nxt.c1.r, nxt.c1.stride, nxt.c1.pad) == (1, 1, 0) and nxt.c1.scale is not None
                and vh.bottleneck_chain_supported(b.c3.w.shape[-1], b.c3.cout, nxt.c1.cout, m))
        y = b.c2(y1 if y1 is not None else b.c1(x, relu=True), relu=True)
        if link:
            x, y1 = vh.bottleneck_chain_fwd(y, b.c3.w, b.c3.scale, b.c3.bias, x, nxt.c1.w, nxt.c1.scale, nxt.c1.bias, out=o)
        else:
            x, y1 = vh.bottleneck_chain_fwd(y, b.c3.w, b.c3.scale, b.c3.bias, x, out=o)[0], None
    return x


# conv1 + bn1 + relu + maxpool of the ResNet trunk as ONE launch that reads the NCHW crops directly (csrc/stem_pool.hip: K = 168 instead of the 224 of the
# 4-channel implicit GEMM, no 3.2 GB stem activation per 1024 crops, no layout pass) where the input size allows (256x192: yes; 384x288: no).  Not in the
# small-batch module calls (vh.latency_mode(): a handful of crops leaves most of its per-image blocks without work).  False = three launches (K summed in another order).
FUSE_STEM = True


def _stem_pool_weight(net):
    c = net.conv1
    ok = (FUSE_STEM and tuple(c.weight.shape) == (64, 3, 7, 7) and c.stride == (2, 2) and c.padding == (3, 3) and c.bias is None
          and isinstance(net.maxpool, nn.MaxPool2d) and net.maxpool.kernel_size in (3, (3, 3)) and net.maxpool.stride in (2, (2, 2))
          and net.maxpool.padding in (1, (1, 1)) and not net.maxpool.ceil_mode)
    return vh.pack_stem_pool_weight(c.weight.detach()) if ok else None


class _TrunkPlan:
    def __init__(self, net):
        self.stem = _Conv(net.conv1, net.bn1)
        self.stem_pw = _stem_pool_weight(net)
        self.blocks = [_BottleneckPlan(b) for stage in net.stages() for b in stage]
        self.n_stage1 = len(net.stages()[0])

    def _stem(self, x_nchw):
        if self.stem_pw is not None and not vh.latency_mode() and vh.stem_pool_supported(x_nchw.shape[2], x_nchw.shape[3]):
            return vh.stem_pool_fwd(x_nchw, self.stem_pw, self.stem.scale, self.stem.bias)
        x = vh.nchw_to_nhwc(x_nchw, 4)                              # 3 -> 4 channels (zero), 16-byte pixels
        x = self.stem(x, relu=True)
        return vh.maxpool3x3s2_fwd(x)

    def _stage1(self, x_nchw, out=None):
        return _run_blocks(self.blocks[:self.n_stage1], self._stem(x_nchw), out=out)

    def __call__(self, x_nchw):
        x = self._stage1(x_nchw)
        for b in self.blocks[self.n_stage1:]:
            x = b(x)
        return x


class _SimplePosePlan:
    def __init__(self, m):
        self.trunk = _TrunkPlan(m.preact)
        d = m.deconv_layers
        self.deconvs = [_Deconv(d[0], d[1]), _Deconv(d[3], d[4]), _Deconv(d[6], d[7])]
        self.head = _Conv(m.final_layer, None)

    def features(self, x_nchw):
        return self.trunk(x_nchw)

    def __call__(self, x_nchw, out=None, emb_out=None):
        x = self.trunk(x_nchw)
        if emb_out is not None:
            emb_out.copy_(vh.gap_fwd(x))                # get_embedding of the same trunk pass (simplepose.py:88-91)
        for dc in self.deconvs:
            x = dc(x)
        return self.head(x, relu=False, out_nchw=True, out=out)


class _Linear:
    """nn.Linear as a 1x1 convolution over a (B,1,1,C) NHWC tensor (same MFMA kernel)."""
    __slots__ = ("w", "scale", "bias", "cout")

    def __init__(self, lin: nn.Linear):
        self.cout, cin = lin.weight.shape
        self.w = vh.pack_conv_weight(lin.weight.detach().reshape(self.cout, cin, 1, 1))
        self.scale, self.bias = vh.bn_fold(None, None, None, None, 0.0, lin.bias.detach(), channels=self.cout)

    def __call__(self, x2d, relu):
        b, c = x2d.shape
        return vh.conv2d_fwd(x2d.reshape(b, 1, 1, c), self.w, self.scale, self.bias, self.cout, 1, 1, 1, 0, relu).reshape(b, self.cout)


class _SEBottleneckPlan(_BottleneckPlan):
    """First block of a FastPose stage: conv3+BN output is gated by sigmoid(fc(avgpool)) before
    the projection shortcut is added (SE_Resnet.py:110-137, SE_module.py:20-24)."""

    def __init__(self, blk):
        super().__init__(blk)
        self.fc1, self.fc2 = _Linear(blk.se.fc[0]), _Linear(blk.se.fc[2])

    def __call__(self, x):
        y = self.c1(x, relu=True)
        y = self.c2(y, relu=True)
        y = self.c3(y, relu=False)
        gate = self.fc2(self.fc1(vh.gap_fwd(y), relu=True), relu=False)     # pre-sigmoid
        return vh.se_scale_add_relu(y, gate, self.proj(x, relu=False))


class _SETrunkPlan(_TrunkPlan):
    def __init__(self, net):
        self.stem = _Conv(net.conv1, net.bn1)
        self.stem_pw = _stem_pool_weight(net)
        self.blocks = [(_SEBottleneckPlan(b) if getattr(b, "reduc", False) else _BottleneckPlan(b))
                       for stage in net.stages() for b in stage]
        self.n_stage1 = len(net.stages()[0])


class _FastPosePlan:
    def __init__(self, m):
        self.trunk = _SETrunkPlan(m.preact)
        self.duc1 = _Conv(m.duc1.conv, m.duc1.bn)
        self.duc2 = _Conv(m.duc2.conv, m.duc2.bn)
        self.head = _Conv(m.conv_out, None)

    def features(self, x_nchw):
        return self.trunk(x_nchw)

    def __call__(self, x_nchw, out=None, emb_out=None):
        t = self.trunk(x_nchw)
        if emb_out is not None:
            emb_out.copy_(vh.gap_fwd(t))                # fastpose.py:70-73
        x = vh.pixelshuffle2_fwd(t)
        x = vh.pixelshuffle2_fwd(self.duc1(x, relu=True))
        x = vh.pixelshuffle2_fwd(self.duc2(x, relu=True))
        return self.head(x, relu=False, out_nchw=True, out=out)


class _BasicBlockPlan:
    def __init__(self, blk):
        self.c1 = _Conv(blk.conv1, blk.bn1)
        self.c2 = _Conv(blk.conv2, blk.bn2)
        self.proj = _Conv(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None

    def __call__(self, x):
        skip = x if self.proj is None else self.proj(x, relu=False)
        return self.c2(self.c1(x, relu=True), relu=True, residual=skip)


def _block_plan(blk):
    return _BottleneckPlan(blk) if hasattr(blk, "conv3") else _BasicBlockPlan(blk)


class _ConvChain:
    """Sequential of Conv-BN(-ReLU) groups (HRNet transitions and strided fusion paths)."""

    def __init__(self, seq):
        groups = [seq] if isinstance(seq[0], nn.Conv2d) else list(seq)
        self.steps = [(_Conv(g[0], g[1]), len(g) > 2 and isinstance(g[2], nn.ReLU)) for g in groups]

    def __call__(self, x, residual=None, final_relu=None):
        for k, (conv, relu) in enumerate(self.steps):
            last = k == len(self.steps) - 1
            x = conv(x, relu=(final_relu if (last and final_relu is not None) else relu), residual=residual if last else None)
        return x


class _HRModulePlan:
    """Branches of BasicBlocks, then per output resolution i:
    y_i = relu( x_i + sum_{j<i} down_ij(x_j) + sum_{j>i} up(conv1x1_ij(x_j)) )   (hrnet.py:242-260).
    Strided paths accumulate through the conv epilogue's residual input; the up-sampled
    terms are added (and the ReLU applied) by one vatl_fuse_upsample_add launch."""

    def __init__(self, mod):
        self.branches = [[_block_plan(b) for b in br] for br in mod.branches]
        self.rows = []
        if mod.fuse_layers is not None:
            for i, row in enumerate(mod.fuse_layers):
                downs = [(j, _ConvChain(row[j])) for j in range(i)]
                ups = [(j, _Conv(row[j][0], row[j][1])) for j in range(i + 1, len(row))]
                self.rows.append((i, downs, ups))

    def __call__(self, xs):
        xs = list(xs)
        for i, br in enumerate(self.branches):
            for blk in br:
                xs[i] = blk(xs[i])
        if not self.rows:
            return xs
        out = []
        for i, downs, ups in self.rows:
            acc = xs[i]
            for k, (j, chain) in enumerate(downs):
                acc = chain(xs[j], residual=acc, final_relu=(not ups and k == len(downs) - 1))
            if ups:
                acc = vh.fuse_upsample_add(acc, [(conv(xs[j], relu=False), j - i) for j, conv in ups], relu=True)
            out.append(acc)
        return out


class _HRNetPlan:
    def __init__(self, m):
        self.stem1, self.stem2 = _Conv(m.conv1, m.bn1), _Conv(m.conv2, m.bn2)
        c = m.conv1
        ok = FUSE_STEM and tuple(c.weight.shape) == (64, 3, 3, 3) and c.stride == (2, 2) and c.padding == (1, 1) and c.bias is None
        self.stem1_pw = vh.pack_stem3_weight(c.weight.detach()) if ok else None
        self.layer1 = [_block_plan(b) for b in m.layer1]
        self.stages = []
        for s in (2, 3, 4):
            trans = [None if t is None else _ConvChain(t) for t in getattr(m, f"transition{s - 1}")]
            self.stages.append((trans, [_HRModulePlan(mod) for mod in getattr(m, f"stage{s}")]))
        self.head = _Conv(m.final_layer, None)

    def __call__(self, x_nchw, out=None):
        if self.stem1_pw is not None and not vh.latency_mode() and vh.stem_pool_supported(x_nchw.shape[2], x_nchw.shape[3]):
            x = vh.stem3_fwd(x_nchw, self.stem1_pw, self.stem1.scale, self.stem1.bias)       # conv1 + bn1 + relu straight from the NCHW crops
        else:
            x = self.stem1(vh.nchw_to_nhwc(x_nchw, 4), relu=True)
        x = _run_blocks(self.layer1, self.stem2(x, relu=True))
        ys = [x]
        for trans, mods in self.stages:
            xs = [ys[i] if t is None else t(ys[-1]) for i, t in enumerate(trans)]
            for mod in mods:
                xs = mod(xs)
            ys = xs
        return self.head(ys[0], relu=False, out_nchw=True, out=out)


class _NchwAdapter:
    """Stand-alone use of a sub-module (ResNet / Bottleneck) on NCHW tensors."""

    def __init__(self, inner, cin_pad=None):
        self.inner, self.cin_pad = inner, cin_pad

    def __call__(self, x_nchw):
        if isinstance(self.inner, _TrunkPlan):
            return vh.nhwc_to_nchw(self.inner(x_nchw))
        return vh.nhwc_to_nchw(self.inner(vh.nchw_to_nhwc(x_nchw)))


def _chunk_limit(hw) -> int:
    """Largest batch one launch sequence may take: MAX_CHUNK, and every tensor below 2^30 elements (the kernels use 32-bit
    buffer byte offsets) — the largest per-crop tensor of all three networks is the stride-2 stem output (H/2 x W/2 x 64):
    1365 crops at 256x192, 606 at 384x288."""
    per_crop = (hw[0] // 2) * (hw[1] // 2) * 64
    return max(1, min(MAX_CHUNK, ((1 << 30) - 1) // max(per_crop, 1)))


def _chunks(n: int, hw=(256, 192)):
    """Balanced chunk bounds: 1080 crops (the reference's evaluation batch) become 2 x 540, not 1024 + 56 — a small last
    chunk would run every layer on a nearly empty grid.  Results do not depend on the chunking (bit-identical)."""
    lim = _chunk_limit(hw)
    k = max(1, -(-n // lim))
    size = -(-n // k)
    return [(i, min(i + size, n)) for i in range(0, n, size)]


def _walk(m: nn.Module):
    """One traversal of the module tree, kept so that the next call only has to CHECK it: (parent, name, child) of every sub-module,
    (owner, kind, name, tensor) of every parameter (kind 0) / buffer (kind 1) in `parameters()` + `buffers()` order (shared tensors once),
    and the sizes of every module's three dicts.  The ROOT module appears as ``None`` everywhere (resolved to the module the walk is
    checked for): the walk is stored in the root's own attribute dict, and a strong reference back to it would be a reference cycle —
    which a host program that calls gc.freeze() never gets rid of (the packed plans and the weights of every model it ever built would
    stay on the device)."""
    mods, links, seen = [m], [], {id(m)}
    k = 0
    while k < len(mods):                                     # breadth first; the ORDER of the signature only has to be stable
        for name, child in mods[k]._modules.items():
            links.append((mods[k] if k else None, name, child))
            if child is not None and id(child) not in seen:
                seen.add(id(child)); mods.append(child)
        k += 1
    tensors, seen_t = [], set()
    for i, mod in enumerate(mods):
        for kind, d in enumerate((mod._parameters, mod._buffers)):
            for name, t in d.items():
                if t is not None and id(t) not in seen_t:
                    seen_t.add(id(t)); tensors.append((mod if i else None, kind, name, t))
    sizes = [(mod if i else None, len(mod._modules), len(mod._parameters), len(mod._buffers)) for i, mod in enumerate(mods)]
    return links, tensors, sizes


def _checked_walk(m: nn.Module):
    """The cached walk of ``m`` if every (parent, name) still holds the same child, every module's CURRENT `_parameters` / `_buffers` dict
    (read from the module now, not remembered) the same tensor objects under the same names and the same number of entries; a fresh walk
    otherwise (a walk recorded for another module object — replicas copy the attribute dict of their original — included)."""
    walk = m.__dict__.get("_vatl_walk")
    if walk is not None and walk[3]() is not m:             # a copy of another module's attribute dict (DataParallel replicas are made that way)
        walk = None
    if walk is not None:
        links, tensors, sizes = walk[:3]
        ok = all((p or m)._modules.get(n) is c for p, n, c in links) and \
            all(((o or m)._buffers if kind else (o or m)._parameters).get(n) is t for o, kind, n, t in tensors) and \
            all(len((mod or m)._modules) == a and len((mod or m)._parameters) == b and len((mod or m)._buffers) == c for mod, a, b, c in sizes)
        if not ok:
            walk = None
    if walk is None:
        walk = m.__dict__["_vatl_walk"] = _walk(m) + (weakref.ref(m),)
    return walk


def _version_key(m: nn.Module, device):
    """Signature of everything a plan bakes in (packed weights, folded BatchNorm): storage address and version counter of
    every parameter and buffer — compared as a tuple, so two different states can never share a key.  Writers that go through
    the C ABI bump the counters themselves (optimizers: active_learning/optim.py; BatchNorm running statistics: hip_train.py).
    `m.parameters()` walks the module tree through Python generators (0.4 ms for ResNet-50, 2 ms for HRNet-W32: a third of a
    single-crop call); the walk is cached and re-validated by identity instead (`_checked_walk`), which catches replaced layers,
    re-assigned parameters and additions alike.  What the key CANNOT see is a write that bumps no counter — `p.data.copy_(w)`,
    `bn.running_var.data.fill_(1)`: `_ParamGuard` below covers those."""
    sig = [str(device)]
    for _, _, _, t in _checked_walk(m)[1]:
        sig.append(t.data_ptr()); sig.append(t._version)
    return tuple(sig)


# PARAMETER GUARD.  True: every plan call is preceded by ONE launch that checksums all parameters and buffers (vatl_checksum_multi: 136 MB
# for SimplePose-R50, 38 us), read back asynchronously and compared with the checksums taken when the plan
# was built; a difference — an in-place write that bumped no version counter — raises StalePlanError naming the tensor at the NEXT call
# into this module (or at `verify(model)`), and the plan is dropped so that the call after the error runs the new values.
# `invalidate(model)` after such a write avoids the error altogether.  False: no guard launches (what round 5 shipped).
PARAM_GUARD = True
# At most this many read-backs in flight per plan: an enqueue-only loop (ActiveLearning.eval_and_query) runs many calls ahead of the
# device; the oldest read-back is waited for once the ring is full.
_GUARD_RING = 8
# ... and at most one guard launch per plan in this many seconds: a stream call (tens of ms) is always guarded; of the ~1 ms module calls of a latency loop
# (`model(x)`, <= 16 crops, ~60 launches) one in twenty is — the guard's ~60 us of stream time are +5 .. +7 % of such a call (tools/latency_bench.py: 0.94 -> 0.98 ms
# at one crop, 1.09 -> 1.17 at four; tools/guard_prof.py: 1.068 -> 1.142 ms guarding every call, 1.086 at this rate).  An untracked write is therefore noticed within one call or this interval,
# whichever is longer; `verify(model)` checks at once.  0 = every call.
GUARD_MIN_INTERVAL_S = 0.02


class StalePlanError(RuntimeError):
    """A parameter or buffer was overwritten in place without torch noticing (`.data` writes bump no version counter) and at least one
    inference call ran on the values packed BEFORE the write.  The plan has been dropped: calls from now on use the new values."""


class _ParamGuard:
    def __init__(self, m: nn.Module, device):
        names = {id(t): n for n, t in list(m.named_parameters()) + list(m.named_buffers())}
        # (tensors the checksum kernel cannot read as aligned 32-bit words — a bool / uint8 buffer of a user's subclass; the three supported networks have none —
        #  are left out rather than failing the plan: `unguarded` names them)
        ok = lambda t: t.device == device and t.numel() and t.is_contiguous() and (t.numel() * t.element_size()) % 4 == 0 and t.data_ptr() % 4 == 0
        walk = [t for _, _, _, t in _checked_walk(m)[1]]
        ts = [t for t in walk if ok(t)]
        self.unguarded = [names.get(id(t), "?") for t in walk if t.numel() and not ok(t)]
        self.names = [names.get(id(t), f"<tensor {i}>") for i, t in enumerate(ts)]
        self.table = vh.ChecksumTable(ts)
        self.device, self.slots, self.last_launch = torch.device(device), None, 0.0
        self.ref = self.table.launch().cpu()                 # plan build is the slow path: a synchronous read-back here is fine
        self.pending = []                                    # [(event, pinned host copy)] in launch order
        self.model = weakref.ref(m)

    def launch(self, force: bool = True):
        now = time.perf_counter()
        if not force and now - self.last_launch < GUARD_MIN_INTERVAL_S:
            return
        self.last_launch = now
        if self.device.type != "cuda":                       # (a host-side stand-in table: tests/test_boundary.py drives this class without a GPU)
            self.pending.append((None, self.table.launch()))
            return
        # On the CALLING stream, in front of the plan's launches (memset + checksum kernel + a 2.7 KB read-back: ~60 us of stream time).  A side stream ordered behind
        # the calling one was measured and rejected (round 6, tools/guard_prof.py): free for a 256-crop call, but every activation of the otherwise idle second
        # queue cost the calling stream ~1 ms — with one guarded call in twenty a 1.07 ms module call still averaged 1.12 ms.
        if self.slots is None:                               # everything a launch needs is made ONCE: a pinned allocation or an event creation per call costs more than the launch
            self.slots = [(torch.empty((self.table.n,), dtype=torch.int64, device=self.device), torch.empty((self.table.n,), dtype=torch.int64, pin_memory=True),
                           torch.cuda.Event()) for _ in range(_GUARD_RING + 2)]
            self.slot = 0
        dev, host, ev = self.slots[self.slot]                # (at most _GUARD_RING read-backs are pending: this slot's last use has been consumed)
        self.slot = (self.slot + 1) % len(self.slots)
        self.table.launch(out=dev)
        host.copy_(dev, non_blocking=True)
        ev.record()
        self.pending.append((ev, host))

    def check(self, wait: bool = False):
        """Compare every finished read-back (all of them with ``wait``; the oldest ones beyond the ring in any case)."""
        while self.pending and (wait or len(self.pending) > _GUARD_RING or self.pending[0][0] is None or self.pending[0][0].query()):
            ev, host = self.pending.pop(0)
            if ev is not None:
                ev.synchronize()
            if not torch.equal(host, self.ref):
                bad = [self.names[i] for i in torch.nonzero(host != self.ref).flatten().tolist()]
                host = None
                self.pending.clear()
                m = self.model()
                if m is not None:
                    invalidate(m)
                raise StalePlanError(
                    f"{len(bad)} parameter / buffer tensor(s) were overwritten in place without a version bump (`.data` write?) after the "
                    f"inference plan was built: {', '.join(bad[:6])}{' ...' if len(bad) > 6 else ''}.  At least one earlier call ran on the OLD "
                    "values; the plan has been dropped and the next call packs the current ones.  Call "
                    "alphapose.models.hip_engine.invalidate(model) after writing through `.data` to avoid this error.")


def invalidate(m: nn.Module) -> None:
    """Forget everything cached for ``m`` (packed filters, folded BatchNorm, the module walk): the next call re-packs from the
    tensors' current values.  The thing to call after writing parameters or buffers in a way torch does not track —
    `p.data.copy_(w)`, `bn.running_mean.data.zero_()`, a raw pointer write — and harmless otherwise."""
    for mod in m.modules():
        mod.__dict__.pop("_vatl_plan", None)
        mod.__dict__.pop("_vatl_walk", None)


def verify(m: nn.Module) -> None:
    """Synchronous form of the parameter guard: waits for every outstanding read-back of ``m``'s plan, checksums the tensors once more NOW
    and raises StalePlanError if anything differs from the plan's packed state.  (No-op for a model without a plan.)"""
    cached = m.__dict__.get("_vatl_plan")
    if cached is None or cached[2] is None:
        return
    cached[2].check(wait=True)
    cached[2].launch()
    cached[2].check(wait=True)


def _plan_for(m: nn.Module, device):
    key = _version_key(m, device)
    x_is_cuda = torch.device(device).type == "cuda"
    cached = m.__dict__.get("_vatl_plan")
    if cached is not None and cached[0] == key:
        if cached[2] is not None and not (x_is_cuda and torch.cuda.is_current_stream_capturing()):
            cached[2].check()                                # read-backs of earlier calls that have arrived
            cached[2].launch(force=False)                    # this call's checksums (side stream), unless the last ones are younger than GUARD_MIN_INTERVAL_S
        return cached[1]
    from .fastpose import FastPose
    from .hrnet import PoseHighResolutionNet
    from .simplepose import SimplePose
    from .layers.Resnet import Bottleneck, ResNet
    from .layers.SE_Resnet import SEResnet
    with torch.no_grad():
        if isinstance(m, SimplePose):
            plan = _SimplePosePlan(m)
        elif isinstance(m, FastPose):
            plan = _FastPosePlan(m)
        elif isinstance(m, PoseHighResolutionNet):
            plan = _HRNetPlan(m)
        elif isinstance(m, SEResnet):
            plan = _NchwAdapter(_SETrunkPlan(m))
        elif isinstance(m, ResNet):
            plan = _NchwAdapter(_TrunkPlan(m))
        elif isinstance(m, Bottleneck):
            plan = _NchwAdapter(_BottleneckPlan(m))
        else:
            raise TypeError(f"no HIP plan for {type(m).__name__}")
        guard = _ParamGuard(m, torch.device(device)) if PARAM_GUARD else None      # AFTER packing: reference = what the packers read
    m.__dict__["_vatl_plan"] = (key, plan, guard)
    return plan


# Module calls `model(x)` with at most this many crops (scripts/poseestimator_eval.py style, BASELINE.json configs[0]: B = 4) run
# with split-K for their duration: every conv launch of such a batch is a fraction of one round of blocks (SimplePose-R50
# forward + decode at B = 4: 2.6 -> 1.2 ms).  Results agree with the unsplit kernels to fp32 rounding, not bit for bit, so the
# evaluation stream of ActiveLearning (forward_into / forward_with_embedding: THC de-duplication needs a crop's bits to be
# independent of its batch) never uses it.  0 switches it off.
SPLITK_AUTO_MAX = 16


def _prepare_input(m: nn.Module, x: torch.Tensor):
    if not x.is_cuda:
        raise vh.VatlError("the pose network runs on MI355X only: move the model and inputs to a HIP device "
                           "(there is deliberately no CPU fallback)")
    p = next(m.parameters())
    if p.device != x.device:
        raise vh.VatlError(f"model on {p.device}, input on {x.device}")
    return x.detach().float().contiguous()


def run_module_nchw(m: nn.Module, x: torch.Tensor) -> torch.Tensor:
    if m.training:                                   # batch-statistics BN + backward tape (hip_train.py)
        from . import hip_train
        return hip_train.forward_train(m, x)
    x = _prepare_input(m, x)
    plan = _plan_for(m, x.device)
    if 0 < x.shape[0] <= min(SPLITK_AUTO_MAX, _chunk_limit(x.shape[2:])):
        with vh.splitk_scope(x.device):
            return plan(x)
    if x.shape[0] <= _chunk_limit(x.shape[2:]):
        return plan(x)
    return torch.cat([plan(x[a:b]) for a, b in _chunks(x.shape[0], x.shape[2:])], 0)


def forward_into(m: nn.Module, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """Heat-maps of ``x`` written straight into ``out`` (N,J,H/4,W/4), a contiguous fp32
    device tensor (e.g. a slice of a whole-video buffer): no concat / copy kernels."""
    if m.training:
        raise vh.VatlError("forward_into is an inference entry point: call model.eval() first")
    x = _prepare_input(m, x)
    plan = _plan_for(m, x.device)
    for a, b in _chunks(x.shape[0], x.shape[2:]):
        plan(x[a:b], out=out[a:b])
    return out


def forward_with_embedding(m: nn.Module, x: torch.Tensor, out: torch.Tensor, emb: torch.Tensor) -> None:
    """Heat-maps into ``out`` and the get_embedding vectors (N, 2048) into ``emb`` from ONE trunk pass — the reference
    runs the trunk twice for this (ActiveLearning.py:277, 284).  Bit-identical to
    forward_into() + embedding(): same launches, the pooled trunk output is simply kept."""
    if m.training:
        raise vh.VatlError("forward_with_embedding is an inference entry point: call model.eval() first")
    x = _prepare_input(m, x)
    plan = _plan_for(m, x.device)
    if not hasattr(plan, "features"):
        raise vh.VatlError(f"{type(m).__name__} has no get_embedding")
    for a, b in _chunks(x.shape[0], x.shape[2:]):
        plan(x[a:b], out=out[a:b], emb_out=emb[a:b])


def embedding(m: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """trunk -> global average pool -> (B, 2048)   (simplepose.py:88-91)."""
    if m.training:
        raise vh.VatlError("get_embedding is used in evaluation only (ActiveLearning.py:259,284): call model.eval() first")
    x = _prepare_input(m, x)
    plan = _plan_for(m, x.device)
    return torch.cat([vh.gap_fwd(plan.features(x[a:b])) for a, b in _chunks(x.shape[0], x.shape[2:])], 0)
