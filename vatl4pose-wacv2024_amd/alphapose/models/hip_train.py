"""Training-mode forward and backward of the pose networks on libvatl_hip.so.

What runs under ``model.train()`` in the reference's fine-tune loop
(ActiveLearning.py:658-673): conv -> BatchNorm with *batch* statistics (+ running-stat
update) -> ReLU, and the matching backward.  Every arithmetic step is a hand-written
gfx950 kernel; torch only owns the buffers and (through one ``autograd.Function`` per
network) receives the parameter gradients, so the reference's
``loss.backward(); optimizer.step()`` works unchanged.

Per Conv+BN(+add)(+ReLU) layer:
  forward   z = conv(x) (fp32 MFMA implicit GEMM, no epilogue affine) ; batch mean/var over
            N*H*W -> (scale, bias), running stats ; y = relu(z*scale + bias (+ skip))
  backward  g = dy*[y>0] ; (dgamma, dbeta) ; dz = gamma*invstd*(g - dbeta/M - xhat*dgamma/M) ;
            dW = wgrad(x, dz) (MFMA, split over pixels) ;
            dx = conv(dz, flipped/transposed W) — for a stride-2 conv as per-parity launches over
            the dz grid (no zero stuffing), the skip gradient rides in as the epilogue residual.
SimplePose (ResNet bottlenecks + deconv head), FastPose (SE bottlenecks, PixelShuffle/DUC head, biased 3x3
output conv) and HRNet (basic blocks, transitions, multi-resolution fusion; trained by the reference with
RETRAIN.OPTIMIZER SGD/Adam) are wired.
"""
from __future__ import annotations

import torch
import torch.nn as nn

import vatl_hip as vh


def _flipped_taps(r, s):
    return [(r - 1 - i, s - 1 - j) for i in range(r) for j in range(s)]


class _Grads(dict):
    """{parameter: gradient} of one backward pass.  With a ``GradArena`` (active_learning/distributed.py) every gradient
    kernel writes straight into the parameter's slice of the flat arena (``out(p)``), and ``flush()`` — called by the
    sequential trainers after every block — lets the arena start the all-reduce of the part that is already final while
    the backward pass goes on.  Without an arena the gradients are fresh tensors."""

    def __init__(self, arena=None, overlap=False):
        super().__init__()
        self.arena, self.overlap = arena, overlap and arena is not None
        self._low, self._filled = (arena.total if arena is not None else 0), 0

    def out(self, p):
        return self.arena.views.get(p) if self.arena is not None else None

    def __setitem__(self, p, t):
        v = self.out(p)
        if v is not None:
            if t.data_ptr() != v.data_ptr():
                v.copy_(t.reshape(v.shape))
            t = v
            if p not in self:
                self._filled += p.numel()
                self._low = min(self._low, self.arena.offset[p])
        super().__setitem__(p, t)

    def close(self):
        """End of a backward pass.  The arena's bucket sequence is fixed (rank-invariant), so a parameter the trainer never wrote
        cannot desynchronise the ranks — but it silently turns the early buckets off (``flush`` needs a gap-free top) and leaves
        a stale slice in the arena: a wiring error, reported where it happens."""
        if self.arena is not None:
            missing = self.arena.total - self._filled
            if missing:
                names = [tuple(p.shape) for p in self.arena.params if p not in self]
                raise RuntimeError(f"backward pass left {missing} gradient elements of the arena unwritten: parameter shapes {names[:8]}")
        return self

    def flush(self):
        # everything at or above the lowest offset written so far is final once that whole range has been written
        if self.overlap and self._filled == self.arena.total - self._low and self.arena.would_fire(self._low):
            _side.join()                                   # the bucket's weight gradients were written on the side stream
            self.arena.done_offset(self._low)


class _SideStream:
    """Weight gradients on a second HIP stream.  In the backward pass of a layer, dW = wgrad(x, dz) and dx = dgrad(dz) both
    consume dz and nothing downstream needs dW before the optimizer step, so the wgrad launches run beside the critical chain
    (BatchNorm backward -> data gradient -> previous layer): at fine-tune batch sizes every conv launch is only 1-3 rounds of
    resident blocks, and the MFMA-bound wgrad blocks fill the tails of the dgrad launches and overlap the HBM-bound BatchNorm
    passes.  Same kernels, same arguments: results are bit-identical to the single-stream order (tests/test_gpu_train_fullsize.py).
    ``_side.enabled = False`` (profiling tools: per-kernel durations then add up to the step time) = one stream."""

    def __init__(self):
        self.enabled = True
        self._streams = {}
        self.used = False

    def stream(self, device):
        st = self._streams.get(device)
        if st is None:
            st = self._streams[device] = torch.cuda.Stream(device=device)
        return st

    def run(self, fn, *tensors):
        """fn() on the side stream once everything queued on the current stream so far is done; ``tensors`` are its inputs
        (kept from being recycled by the caching allocator until the side stream is through with them)."""
        if not self.enabled:
            return fn()
        main = torch.cuda.current_stream()
        side = self.stream(tensors[0].device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out = fn()
        for t in tensors:
            t.record_stream(side)
        self.used = True
        return out

    def join(self):
        """The current stream waits for every side-stream launch so far (before gradients are read / all-reduced)."""
        if self.used:
            for st in self._streams.values():
                torch.cuda.current_stream().wait_stream(st)
            self.used = False


_side = _SideStream()


def _gout(grads, p):
    """The arena slice a gradient kernel should write for parameter p (None: let the wrapper allocate; plain dicts are accepted
    wherever a ``_Grads`` is)."""
    return grads.out(p) if isinstance(grads, _Grads) else None


# ROUTE CONSTANTS of the trainers (module attributes, not environment switches; tests / tools that A/B a route set them):
# BatchNorm-backward reduction inside the producing data-gradient launch (vatl_conv2d_fwd_ex_bnbwd); False = the
# stand-alone reduction pass (same values up to the summation order of the per-channel sums)
_FUSE_BN_BWD = True
_WINOGRAD = True    # 3x3 / stride-1 layers: forward + data gradient as Winograd F(2x2,3x3)
# ... or as F(4x4,3x3) (csrc/winograd_f4.hip: 0.5625x the multiplies of F(2x2); statistics and BatchNorm-backward epilogues built and tested,
# tests/test_gpu_winograd.py::test_winograd_f4_training_epilogues) where the grid is whole 4x4 tiles and the channel counts fit its 64-channel blocks (ResNet stages 1 - 3 at
# 256x192, stages 1 - 2 at 384x288).  OFF for the fine-tune step, on measurement (profiles/r05_notes.md): at B = 120 / 32 the step gains 0.8 % / 1.5 % (36.8 -> 36.5 ms,
# 55.1 -> 54.3 ms: 22 of ~500 launches, one more re-pack each), while F(4x4)'s larger transform constants cost ~4 bits per layer — invisible in the heat-maps, but in the
# ill-conditioned default-initialisation gradient fixture (tests/test_gpu_train.py::test_b16_default_init_step_vs_reference_and_float64) the trunk gradients land 1.5 - 1.7x
# as far from float64 as torch's own fp32 step (3.4e-2 against 2.1e-2), over that test's "as close to float64 as the reference is" bound.  Not worth it for < 2 %.
_WINOGRAD_F4 = False
# ... and the weight gradient from this many channels on (measured at B = 120, tools/wino_wgrad_bench.py: 128 channels 1.30x, 256 1.35x, 512 1.44x
# over the implicit GEMM; 64 channels 1.0x, 32 channels slower — both operands are transformed per tile pair, 2.5x the vector work of the forward)
_WINOGRAD_WGRAD_MIN_C = 128
# ... for layers with at least this many channels: on the narrow tiles (32 / 64 output channels) the statistics epilogue costs
# more than the stand-alone reduction pass it replaces (HRNet-W32 step 63.9 -> 69.4 ms with every layer fused)
_FUSE_BN_MIN_C = 128

# tests only: when a list, every ReLU the trainers apply appends its output's mask (y > 0), in execution order — what lets a float64
# reference of a block be evaluated with exactly the ReLU decisions the fp32 forward took (tests/test_gpu_train.py)
_relu_tap = None


def _tap(y):
    if _relu_tap is not None:
        _relu_tap.append(y > 0)
    return y


_pending_counters = []


def _count_batch(bn):
    """``num_batches_tracked += 1`` of a BatchNorm layer, deferred: the trainers bump all counters of a forward pass with
    one multi-tensor add (a 53- to 292-layer network otherwise spends a launch per layer on a one-element add)."""
    _pending_counters.append(bn.num_batches_tracked)
    _pending_stats.append(bn.running_mean)
    _pending_stats.append(bn.running_var)


_pending_stats = []


def _flush_batch_counters():
    if _pending_counters:
        torch._foreach_add_(_pending_counters, 1)
        _pending_counters.clear()
    # the running statistics were updated through raw pointers (vatl_bn_train_finalize): bump their version counters so that the
    # inference plans (which fold them into conv epilogues, hip_engine._version_key) see the change
    for t in _pending_stats:
        torch.autograd.graph.increment_version(t)
    _pending_stats.clear()


_USE_PACK_PLAN = True      # False: every re-pack its own launch (same bits; tests/test_gpu_train.py A/Bs it)


def _planned(kind):
    """Trainer.forward / Trainer.backward under the trainer's PackPlan (vatl_hip.PackPlan): the forward pass starts with ONE
    launch that refreshes every packed weight copy the step will use (forward layouts and data-gradient layouts), the pack calls
    inside both passes then return the kept buffers; the first step records what is needed, the backward pass seals the record."""
    def deco(fn):
        def inner(self, *a, **k):
            if not _USE_PACK_PLAN:
                with vh.streamk_scope(a[0].device):
                    return fn(self, *a, **k)
            plan = self.__dict__.get("_pack_plan")
            if plan is None:
                plan = self.__dict__["_pack_plan"] = vh.PackPlan()
            prev = vh.set_pack_plan(plan)
            try:
                if kind == "forward":
                    plan.begin()
                with vh.streamk_scope(a[0].device):            # small launches share their (tile, k-tile) units over the whole chip
                    out = fn(self, *a, **k)
                if kind == "backward":
                    plan.seal()
                return out
            finally:
                vh.set_pack_plan(prev)
        inner.__name__, inner.__doc__ = fn.__name__, fn.__doc__
        return inner
    return deco


def _wino_wgrad_fits(xshape, mo: int) -> bool:
    """winograd_wgrad_impl addresses tiles with 20 bits (csrc/winograd_wgrad.hip: N * ceil(H / MO) * ceil(W / MO) < 2^20, N * tile rows < 2^20):
    batches beyond that (> 1365 crops at 64 x 48) take the implicit-GEMM weight gradient instead of failing mid-step.  (The forward / data
    gradient kernels share their only limit, 2^30 elements per tensor, with the implicit GEMM: nothing to fall back to there.)"""
    n, h, w = int(xshape[0]), int(xshape[1]), int(xshape[2])
    th, tw = -(-h // mo), -(-w // mo)
    return n * th * tw < (1 << 20) and n * th < (1 << 20)


class _ConvBN:
    """Conv2d (bias-free) + BatchNorm2d(train) (+ residual) (+ ReLU)."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, relu: bool, need_dx: bool = True):
        self.conv, self.bn, self.relu, self.need_dx = conv, bn, relu, need_dx
        self.cout, self.cin, self.r, self.s = conv.weight.shape
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        # 3x3 / stride 1 / pad 1: forward and data gradient on the Winograd route (csrc/conv_winograd.hip; geometry-only choice)
        self.wino = (_WINOGRAD and (self.r, self.s, self.stride, self.pad) == (3, 3, 1, 1) and self.cin % 16 == 0 and self.cout % 16 == 0)

    @staticmethod
    def _f4(shape, cin, cout):
        """F(4x4,3x3) serves this launch: whole 4x4 tiles, >= 64 input channels (a multiple of 16), output channels a multiple of 64."""
        return _WINOGRAD_F4 and cin >= 64 and cin % 16 == 0 and cout % 64 == 0 and vh.conv3x3_winograd_f4_supported(int(shape[0]), int(shape[1]), int(shape[2]), cin, cout)

    # ---- forward -------------------------------------------------------------
    def forward(self, x, skip=None, relu=None):
        relu = self.relu if relu is None else relu
        bn = self.bn
        # z = conv(x); the batch statistics come out of the conv epilogue (no extra pass over z)
        if self.wino and self._f4(x.shape, self.cin, self.cout):
            z, mean, invstd, scale, bias = vh.conv3x3_winograd_f4_fwd_bnstats(x, vh.pack_winograd_f4_weight(self.conv.weight.detach()), self.cout,
                                                                              bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                                              bn.momentum, bn.eps)
        elif self.wino:
            z, mean, invstd, scale, bias = vh.conv3x3_winograd_fwd_bnstats(x, vh.pack_winograd_weight(self.conv.weight.detach()), self.cout,
                                                                           bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                                           bn.momentum, bn.eps)
        else:
            w = vh.pack_conv_weight(self.conv.weight.detach())
            z, mean, invstd, scale, bias = vh.conv2d_fwd_bnstats(x, w, self.cout, self.r, self.s, self.stride, self.pad, bn.weight.detach(),
                                                                 bn.bias.detach(), bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        _count_batch(bn)
        y = vh.scale_bias_act(z, scale, bias, skip, relu)
        if relu:
            _tap(y)
        # ReLU without a skip: the backward recomputes the mask from (z, scale, bias) and never reads y
        mask = (scale, bias) if (relu and skip is None) else None
        self.saved = (x, z, y if (relu and skip is not None) else None, mean, invstd, skip is not None, mask)
        return y

    def forward_pool(self, x):
        """Conv + BN(train) + ReLU + MaxPool2d(3,2,1) (the trunk's stem tail, Resnet.py:171-172): the affine + ReLU are applied to z
        inside the pooling pass, relu(bn(z)) itself is never written.  -> pooled output; the winners stay on the tape."""
        w = vh.pack_conv_weight(self.conv.weight.detach())
        bn = self.bn
        z, mean, invstd, scale, bias = vh.conv2d_fwd_bnstats(x, w, self.cout, self.r, self.s, self.stride, self.pad, bn.weight.detach(),
                                                             bn.bias.detach(), bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        _count_batch(bn)
        y, idx = vh.maxpool3x3s2_fwd_idx_affine(z, scale, bias)
        self.saved = (x, z, None, mean, invstd, False, (scale, bias))
        self.pool_idx = idx
        return y

    def backward_pool(self, dpool, grads):
        """Backward of forward_pool from the pooled output's gradient (no full-resolution gradient tensor).  The stem's own input
        gradient is never needed (need_dx False)."""
        x, z, _, mean, invstd, _, mask = self.saved
        self.saved = None
        idx, self.pool_idx = self.pool_idx, None
        dz, dgamma, dbeta = vh.bn_train_bwd_relu_pool(dpool, idx, mask[0], mask[1], z, self.bn.weight.detach(), mean, invstd,
                                                      dgamma=_gout(grads, self.bn.weight), dbeta=_gout(grads, self.bn.bias))
        grads[self.bn.weight] = dgamma
        grads[self.bn.bias] = dbeta
        cin_w = 3 if self.cin == 3 else self.cin
        ow = _gout(grads, self.conv.weight)
        grads[self.conv.weight] = _side.run(lambda: vh.conv2d_wgrad(x, dz, self.cout, cin_w, self.r, self.s, self.stride, self.pad, out=ow), x, dz)
        assert not self.need_dx

    # ---- backward ------------------------------------------------------------
    def bn_spec(self):
        """What the launch that PRODUCES this layer's output gradient needs in order to run the reduction pass of this layer's
        BatchNorm backward in its own epilogue (vatl_conv2d_fwd_ex_bnbwd); taken before ``backward`` consumes the tape."""
        if not _FUSE_BN_BWD or self.cout < _FUSE_BN_MIN_C:
            return None
        x, z, y, mean, invstd, had_skip, mask = self.saved
        if mask is not None:                               # ReLU, no skip: mask recomputed from z
            return vh.BnBwdSpec(z, mean, invstd, scale=mask[0], bias=mask[1])
        return vh.BnBwdSpec(z, mean, invstd, mask_y=y)     # ReLU after a skip sum (y saved), or no ReLU at all (y is None)

    def backward(self, dy, grads, dx_residual=None, pre=None, consumer=None):
        """dy: gradient of the layer output.  Returns (dx, g_skip); parameter gradients go to ``grads``.
        ``pre``: the BnBwdSpec this layer handed to the producer of dy — dy is then already masked and its (sum g, sum g*xhat)
        partials are in ``pre``.  ``consumer``: the BnBwdSpec of the layer that will receive dx."""
        x, z, y, mean, invstd, had_skip, mask = self.saved
        self.saved = None
        og, ob = _gout(grads, self.bn.weight), _gout(grads, self.bn.bias)
        if pre is not None:
            dz, dgamma, dbeta = vh.bn_bwd_from_stats(pre, dy, self.bn.weight.detach(), dgamma=og, dbeta=ob)
            g = dy if had_skip else None
        elif mask is not None:
            dz, dgamma, dbeta = vh.bn_train_bwd_relu(dy, mask[0], mask[1], z, self.bn.weight.detach(), mean, invstd, dgamma=og, dbeta=ob)
            g = None
        else:
            dz, g, dgamma, dbeta = vh.bn_train_bwd(dy, y, z, self.bn.weight.detach(), mean, invstd, want_g=had_skip and y is not None,
                                                   dgamma=og, dbeta=ob)
        if had_skip and y is None:
            g = dy                                         # no ReLU between the sum and the output: the skip gradient is dy
        grads[self.bn.weight] = dgamma
        grads[self.bn.bias] = dbeta
        cin_w = 3 if self.cin == 3 else self.cin
        ow = _gout(grads, self.conv.weight)
        if self.wino and min(self.cin, self.cout) >= _WINOGRAD_WGRAD_MIN_C and _wino_wgrad_fits(x.shape, 2):      # transform-domain weight gradient (csrc/winograd_wgrad.hip)
            grads[self.conv.weight] = _side.run(lambda: vh.conv3x3_winograd_wgrad(x, dz, out=ow), x, dz)
        else:
            grads[self.conv.weight] = _side.run(lambda: vh.conv2d_wgrad(x, dz, self.cout, cin_w, self.r, self.s, self.stride, self.pad, out=ow), x, dz)
        dx = self._dgrad(dz, x.shape, dx_residual, consumer) if self.need_dx else None
        return dx, g

    def _dgrad(self, dz, xshape, residual, spec=None):
        n, h, w, cin = xshape
        wt = self.conv.weight.detach()
        ho, wo = dz.shape[1], dz.shape[2]

        def conv(*a, **k):                                # with a consumer spec: mask + BatchNorm-backward reduction in the epilogue
            return vh.conv2d_fwd_ex_bnbwd(*a, spec, **k) if spec is not None else vh.conv2d_fwd_ex(*a, **k)
        if self.wino and self._f4(dz.shape, self.cout, cin):           # the data gradient is a 3x3 conv of dz with cout input and cin output channels
            ud = vh.pack_winograd_f4_weight(wt, data_gradient=True)
            if spec is not None:
                return vh.conv3x3_winograd_f4_fwd_bnbwd(dz, ud, cin, spec, residual=residual)
            return vh.conv3x3_winograd_f4_fwd(dz, ud, None, None, cin, False, residual=residual)
        if self.wino:
            ud = vh.pack_winograd_weight(wt, data_gradient=True)
            if spec is not None:
                return vh.conv3x3_winograd_fwd_bnbwd(dz, ud, cin, spec, residual=residual)
            return vh.conv3x3_winograd_fwd(dz, ud, None, None, cin, False, residual=residual)
        if self.stride == 1:
            wd = vh.pack_dgrad_weight(wt, _flipped_taps(self.r, self.s))
            return conv(dz, wd, cin, self.r, self.s, 1, self.r - 1 - self.pad, self.s - 1 - self.pad, h, w, h, w, 1, 1, 0, 0, residual=residual)
        assert self.stride == 2 and h == 2 * ho and w == 2 * wo
        if self.r == 1:                                   # 1x1/2 projection: only even pixels receive gradient
            assert spec is None
            dx = torch.zeros((n, h, w, cin), device=dz.device, dtype=torch.float32) if residual is None else residual.clone()
            wd = vh.pack_dgrad_weight(wt, [(0, 0)])
            res_view = None
            if residual is not None:
                # even pixels: dx = W^T dz + residual; the kernel reads the residual at the scattered positions
                res_view = residual
            return vh.conv2d_fwd_ex(dz, wd, cin, 1, 1, 1, 0, 0, ho, wo, h, w, 2, 2, 0, 0, out=dx, residual=res_view)
        assert self.r == 3 and self.pad == 1
        dx = torch.empty((n, h, w, cin), device=dz.device, dtype=torch.float32)
        for py in (0, 1):                                 # input-pixel parity -> which filter rows reach it
            rows = [1] if py == 0 else [2, 0]             # dz row y+t  <->  filter row rows[t]
            for px in (0, 1):
                cols = [1] if px == 0 else [2, 0]
                wd = vh.pack_dgrad_weight(wt, [(a, b) for a in rows for b in cols])
                conv(dz, wd, cin, len(rows), len(cols), 1, 0, 0, ho, wo, h, w, 2, 2, py, px, out=dx, residual=residual)
        return dx


class _DeconvBN:
    """ConvTranspose2d(4,2,1, bias-free) + BatchNorm2d(train) + ReLU."""

    def __init__(self, dc: nn.ConvTranspose2d, bn: nn.BatchNorm2d):
        self.dc, self.bn = dc, bn
        self.cin, self.cout = dc.weight.shape[:2]

    def forward(self, x):
        bn = self.bn
        if _WINOGRAD and self.cin % 16 == 0 and self.cout % 4 == 0:      # four 2x2 phase convolutions as Winograd F(3x3, 2x2)
            z, mean, invstd, scale, bias = vh.deconv4x4s2_winograd_fwd_bnstats(x, vh.pack_winograd_deconv_weight(self.dc.weight.detach()), self.cout,
                                                                               bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                                               bn.momentum, bn.eps)
        else:
            z, mean, invstd, scale, bias = vh.deconv4x4s2_fwd_bnstats(x, vh.pack_deconv_weight(self.dc.weight.detach()), self.cout, bn.weight.detach(),
                                                                      bn.bias.detach(), bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        _count_batch(bn)
        y = _tap(vh.scale_bias_act(z, scale, bias, None, True))
        self.saved = (x, z, scale, bias, mean, invstd)
        return y

    def bn_spec(self):
        if not _FUSE_BN_BWD or self.cout < _FUSE_BN_MIN_C:
            return None
        x, z, scale, bias, mean, invstd = self.saved
        return vh.BnBwdSpec(z, mean, invstd, scale=scale, bias=bias)

    def backward(self, dy, grads, pre=None, consumer=None):
        x, z, scale, bias, mean, invstd = self.saved
        self.saved = None
        if pre is not None:
            dz, dgamma, dbeta = vh.bn_bwd_from_stats(pre, dy, self.bn.weight.detach(), dgamma=_gout(grads, self.bn.weight), dbeta=_gout(grads, self.bn.bias))
        else:
            dz, dgamma, dbeta = vh.bn_train_bwd_relu(dy, scale, bias, z, self.bn.weight.detach(), mean, invstd,
                                                     dgamma=_gout(grads, self.bn.weight), dbeta=_gout(grads, self.bn.bias))
        grads[self.bn.weight] = dgamma
        grads[self.bn.bias] = dbeta
        ow = _gout(grads, self.dc.weight)
        if _WINOGRAD and min(self.cin, self.cout) >= _WINOGRAD_WGRAD_MIN_C and _wino_wgrad_fits(x.shape, 3):      # transform-domain weight gradient (1.04 - 1.16x at B = 120)
            grads[self.dc.weight] = _side.run(lambda: vh.deconv4x4s2_winograd_wgrad(x, dz, out=ow), x, dz)
        else:
            grads[self.dc.weight] = _side.run(lambda: vh.deconv4x4s2_wgrad(x, dz, out=ow), x, dz)
        # dx[y][x][ci] = sum_{ky,kx,co} dz[2y-1+ky][2x-1+kx][co] * W[ci][co][ky][kx]: a 4x4/2 pad-1 conv whose
        # "OIHW" weight is the deconv weight itself (O = Cin, I = Cout)
        if _WINOGRAD and self.cout % 16 == 0 and self.cin % 4 == 0:
            # ... = the sum over the four pixel phases of dz of 2x2 convolutions: Winograd F(3x3,2x2) with the reduction over (phase, channel)
            return vh.deconv4x4s2_winograd_dgrad(dz, vh.pack_winograd_deconv_dgrad_weight(self.dc.weight.detach()), self.cin, spec=consumer)
        wd = vh.pack_conv_weight(self.dc.weight.detach())
        if consumer is not None:
            n, h, w, _ = x.shape
            return vh.conv2d_fwd_ex_bnbwd(dz, wd, self.cin, 4, 4, 2, 1, 1, h, w, h, w, 1, 1, 0, 0, consumer)
        return vh.conv2d_fwd(dz, wd, None, None, self.cin, 4, 4, 2, 1, False)


class _BottleneckT:
    def __init__(self, blk):
        self.c1 = _ConvBN(blk.conv1, blk.bn1, True)
        self.c2 = _ConvBN(blk.conv2, blk.bn2, True)
        self.c3 = _ConvBN(blk.conv3, blk.bn3, True)
        self.proj = _ConvBN(blk.downsample[0], blk.downsample[1], False) if blk.downsample is not None else None

    def forward(self, x):
        skip = x if self.proj is None else self.proj.forward(x)
        return self.c3.forward(self.c2.forward(self.c1.forward(x)), skip=skip)

    def out_spec(self):
        """BnBwdSpec of the block output (= conv3's BatchNorm; the ReLU mask comes from the saved y)."""
        return self.c3.bn_spec()

    def backward(self, dy, grads, pre=None, consumer=None):
        s2, s1 = self.c2.bn_spec(), self.c1.bn_spec()
        db, g = self.c3.backward(dy, grads, pre=pre, consumer=s2)        # g = gradient of the skip input (masked dy)
        da, _ = self.c2.backward(db, grads, pre=s2, consumer=s1)
        dskip = g if self.proj is None else self.proj.backward(g, grads)[0]
        dx, _ = self.c1.backward(da, grads, dx_residual=dskip, pre=s1, consumer=consumer)   # dx = dgrad(c1) + dskip in one epilogue
        return dx


def _chain_blocks_backward(blocks, dx, grads, pre, flush=None):
    """Backward through a run of residual blocks, last to first.  The data-gradient launch that produces a block's input
    gradient runs the reduction pass of the PREVIOUS block's output BatchNorm in its epilogue (``out_spec``) whenever that
    block is a plain residual block; ``pre`` is the spec the caller already filled for the last block (or None)."""
    for i in range(len(blocks) - 1, -1, -1):
        if hasattr(blocks[i], "out_spec"):
            consumer = blocks[i - 1].out_spec() if (i > 0 and hasattr(blocks[i - 1], "out_spec")) else None
            dx = blocks[i].backward(dx, grads, pre=pre, consumer=consumer)
            pre = consumer
        else:                                              # SE bottleneck: gated output, its own backward kernels; plain gradient out
            assert pre is None
            dx = blocks[i].backward(dx, grads)
        if flush is not None:
            flush()
    return dx


class SimplePoseTrainer:
    """Tape-based forward/backward of SimplePose in training mode."""

    def __init__(self, m):
        t = m.preact
        self.stem = _ConvBN(t.conv1, t.bn1, True, need_dx=False)   # the input gradient is never read
        self.blocks = [_BottleneckT(b) for stage in t.stages() for b in stage]
        d = m.deconv_layers
        self.deconvs = [_DeconvBN(d[0], d[1]), _DeconvBN(d[3], d[4]), _DeconvBN(d[6], d[7])]
        self.head = m.final_layer

    @_planned("forward")
    def forward(self, x_nchw):
        x = vh.nchw_to_nhwc(x_nchw, 4)
        x = self.stem.forward_pool(x)                                           # conv + bn + relu + maxpool: the activation is never stored
        for b in self.blocks:
            x = b.forward(x)
        for d in self.deconvs:
            x = d.forward(x)
        self.head_in = x
        hw = vh.pack_conv_weight(self.head.weight.detach())
        _, hb = vh.bn_fold(None, None, None, None, 0.0, self.head.bias.detach(), channels=self.head.weight.shape[0])
        _flush_batch_counters()
        return vh.conv2d_fwd(x, hw, None, hb, self.head.weight.shape[0], 1, 1, 1, 0, False, out_nchw=True)

    @_planned("backward")
    def backward(self, dout_nchw, arena=None, overlap=False):
        """dout (B,J,H,W) NCHW -> {parameter: gradient} for every parameter of the model.  With ``arena`` the gradients are
        its slices; ``overlap`` lets the arena all-reduce finished buckets while the earlier layers are still in flight."""
        grads = _Grads(arena, overlap)
        j = self.head.weight.shape[0]
        cin = self.head.weight.shape[1]
        dy = vh.nchw_to_nhwc(dout_nchw.contiguous(), 32)                         # 17 -> 32 channels (zeros)
        grads[self.head.bias] = vh.col_sum(dy)[:j].contiguous()
        grads[self.head.weight] = vh.conv2d_wgrad(self.head_in, dy, j, cin, 1, 1, 1, 0, out=_gout(grads, self.head.weight))
        wd = vh.pack_dgrad_weight(self.head.weight.detach(), [(0, 0)], cout_k=32)
        n, h, w, _ = self.head_in.shape
        pre = self.deconvs[-1].bn_spec()
        dx = vh.conv2d_fwd_ex_bnbwd(dy, wd, cin, 1, 1, 1, 0, 0, h, w, h, w, 1, 1, 0, 0, pre) if pre is not None else \
            vh.conv2d_fwd_ex(dy, wd, cin, 1, 1, 1, 0, 0, h, w, h, w, 1, 1, 0, 0)
        self.head_in = None
        for k in range(len(self.deconvs) - 1, -1, -1):
            consumer = self.deconvs[k - 1].bn_spec() if k > 0 else self.blocks[-1].out_spec()
            dx = self.deconvs[k].backward(dx, grads, pre=pre, consumer=consumer)
            pre = consumer
            grads.flush()
        dx = _chain_blocks_backward(self.blocks, dx, grads, pre, grads.flush)
        self.stem.backward_pool(dx, grads)
        _side.join()
        return grads.close()


class _LinearT:
    """nn.Linear (+ReLU) on (B,C): the 1x1-conv kernel forward, transposed-weight conv / MFMA wgrad / column sum backward."""

    def __init__(self, lin: nn.Linear, relu: bool):
        self.lin, self.relu = lin, relu
        self.co, self.ci = lin.weight.shape

    def forward(self, x2d):
        b = x2d.shape[0]
        w4 = self.lin.weight.detach().reshape(self.co, self.ci, 1, 1)
        _, bias = vh.bn_fold(None, None, None, None, 0.0, self.lin.bias.detach(), channels=self.co)
        y = vh.conv2d_fwd(x2d.reshape(b, 1, 1, self.ci), vh.pack_conv_weight(w4), None, bias, self.co, 1, 1, 1, 0, self.relu).reshape(b, self.co)
        if self.relu:
            _tap(y)
        self.saved = (x2d, y)
        return y

    def backward(self, dy, grads):
        x2d, y = self.saved
        self.saved = None
        b = x2d.shape[0]
        if self.relu:
            dy = vh.relu_bwd(dy.contiguous(), y)
        grads[self.lin.bias] = vh.col_sum(dy)
        ow = _gout(grads, self.lin.weight)
        grads[self.lin.weight] = vh.conv2d_wgrad(x2d.reshape(b, 1, 1, self.ci), dy.reshape(b, 1, 1, self.co), self.co, self.ci, 1, 1, 1, 0,
                                                 out=None if ow is None else ow.view(self.co, self.ci, 1, 1)).reshape(self.co, self.ci)
        wd = vh.pack_dgrad_weight(self.lin.weight.detach().reshape(self.co, self.ci, 1, 1), [(0, 0)])
        return vh.conv2d_fwd_ex(dy.reshape(b, 1, 1, self.co), wd, self.ci, 1, 1, 1, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0).reshape(b, self.ci)


class _SEBottleneckT:
    """First block of a FastPose stage: y = relu(bn3(conv3(..)) * sigmoid(fc(avgpool(.))) + projection(x))
    (SE_Resnet.py:110-137, SE_module.py:20-24)."""

    def __init__(self, blk):
        self.c1 = _ConvBN(blk.conv1, blk.bn1, True)
        self.c2 = _ConvBN(blk.conv2, blk.bn2, True)
        self.c3 = _ConvBN(blk.conv3, blk.bn3, False)
        self.proj = _ConvBN(blk.downsample[0], blk.downsample[1], False)
        self.fc1, self.fc2 = _LinearT(blk.se.fc[0], True), _LinearT(blk.se.fc[2], False)

    def forward(self, x):
        u = self.c3.forward(self.c2.forward(self.c1.forward(x)))
        gate = self.fc2.forward(self.fc1.forward(vh.gap_fwd(u)))              # pre-sigmoid
        y = _tap(vh.se_scale_add_relu(u, gate, self.proj.forward(x)))
        self.saved = (u, gate, y)
        return y

    def backward(self, dy, grads):
        u, gate, y = self.saved
        self.saved = None
        dgate = vh.se_bwd_gate(dy, y, u, gate)
        dpool = self.fc1.backward(self.fc2.backward(dgate, grads), grads)
        du, gm = vh.se_bwd_apply(dy, y, gate, dpool)
        s2, s1 = self.c2.bn_spec(), self.c1.bn_spec()
        db, _ = self.c3.backward(du, grads, consumer=s2)
        da, _ = self.c2.backward(db, grads, pre=s2, consumer=s1)
        dskip, _ = self.proj.backward(gm, grads)
        dx, _ = self.c1.backward(da, grads, dx_residual=dskip, pre=s1)
        return dx


class FastPoseTrainer:
    """Tape-based forward/backward of FastPose in training mode (fastpose.py:52-59)."""

    def __init__(self, m):
        t = m.preact
        self.stem = _ConvBN(t.conv1, t.bn1, True, need_dx=False)
        self.blocks = [(_SEBottleneckT(b) if getattr(b, "reduc", False) else _BottleneckT(b)) for stage in t.stages() for b in stage]
        self.duc1 = _ConvBN(m.duc1.conv, m.duc1.bn, True)
        self.duc2 = _ConvBN(m.duc2.conv, m.duc2.bn, True)
        self.head = m.conv_out

    @_planned("forward")
    def forward(self, x_nchw):
        x = self.stem.forward_pool(vh.nchw_to_nhwc(x_nchw, 4))                  # conv + bn + relu + maxpool: the activation is never stored
        for b in self.blocks:
            x = b.forward(x)
        x = vh.pixelshuffle2_fwd(x)
        x = vh.pixelshuffle2_fwd(self.duc1.forward(x))
        x = vh.pixelshuffle2_fwd(self.duc2.forward(x))
        self.head_in = x
        j = self.head.weight.shape[0]
        _, hb = vh.bn_fold(None, None, None, None, 0.0, self.head.bias.detach(), channels=j)
        _flush_batch_counters()
        return vh.conv2d_fwd(x, vh.pack_conv_weight(self.head.weight.detach()), None, hb, j, 3, 3, 1, 1, False, out_nchw=True)

    @_planned("backward")
    def backward(self, dout_nchw, arena=None, overlap=False):
        grads = _Grads(arena, overlap)
        j, cin = self.head.weight.shape[:2]
        dy = vh.nchw_to_nhwc(dout_nchw.contiguous(), 32)
        grads[self.head.bias] = vh.col_sum(dy)[:j].contiguous()
        grads[self.head.weight] = vh.conv2d_wgrad(self.head_in, dy, j, cin, 3, 3, 1, 1, out=_gout(grads, self.head.weight))
        wd = vh.pack_dgrad_weight(self.head.weight.detach(), _flipped_taps(3, 3), cout_k=32)
        n, h, w, _ = self.head_in.shape
        dx = vh.conv2d_fwd_ex(dy, wd, cin, 3, 3, 1, 1, 1, h, w, h, w, 1, 1, 0, 0)
        self.head_in = None
        dx, _ = self.duc2.backward(vh.pixelunshuffle2(dx), grads)
        dx, _ = self.duc1.backward(vh.pixelunshuffle2(dx), grads)
        grads.flush()
        dx = vh.pixelunshuffle2(dx)
        dx = _chain_blocks_backward(self.blocks, dx, grads, None, grads.flush)
        self.stem.backward_pool(dx, grads)
        _side.join()
        return grads.close()


class _BasicBlockT:
    """hrnet.py:24-56: relu(bn2(conv2(relu(bn1(conv1(x))))) + skip)."""

    def __init__(self, blk):
        self.c1 = _ConvBN(blk.conv1, blk.bn1, True)
        self.c2 = _ConvBN(blk.conv2, blk.bn2, True)
        self.proj = _ConvBN(blk.downsample[0], blk.downsample[1], False) if blk.downsample is not None else None

    def forward(self, x):
        skip = x if self.proj is None else self.proj.forward(x)
        return self.c2.forward(self.c1.forward(x), skip=skip)

    def out_spec(self):
        return self.c2.bn_spec()

    def backward(self, dy, grads, pre=None, consumer=None):
        s1 = self.c1.bn_spec()
        da, g = self.c2.backward(dy, grads, pre=pre, consumer=s1)
        dskip = g if self.proj is None else self.proj.backward(g, grads)[0]
        return self.c1.backward(da, grads, dx_residual=dskip, pre=s1, consumer=consumer)[0]


def _block_t(blk):
    return _BottleneckT(blk) if hasattr(blk, "conv3") else _BasicBlockT(blk)


class _ChainT:
    """Sequential of Conv-BN(-ReLU) groups: HRNet transitions and the strided paths of a fusion row."""

    def __init__(self, seq):
        groups = [seq] if isinstance(seq[0], nn.Conv2d) else list(seq)
        self.steps = [_ConvBN(g[0], g[1], len(g) > 2 and isinstance(g[2], nn.ReLU)) for g in groups]

    def forward(self, x, skip=None):
        for k, st in enumerate(self.steps):
            x = st.forward(x, skip=skip if k == len(self.steps) - 1 else None)
        return x

    def backward(self, dy, grads, dx_residual=None):
        for k in range(len(self.steps) - 1, -1, -1):
            dy = self.steps[k].backward(dy, grads, dx_residual=dx_residual if k == 0 else None)[0]
        return dy


class _HRModuleT:
    """HighResolutionModule (hrnet.py:242-260): y_i = relu(x_i + sum_{j<i} down_ij(x_j) + sum_{j>i} up(bn(conv1x1_ij(x_j)))).
    Forward: strided paths accumulate through the skip input of their last BN, the up-sampled terms and the ReLU
    are one vatl_fuse_upsample_add launch.  Backward: g_i = dy_i*[y_i>0] goes to x_i unchanged, to the strided
    paths as is and to the 1x1 paths as 2^s x 2^s block sums; the per-branch sums ride on the dgrad epilogues."""

    def __init__(self, mod):
        self.branches = [[_block_t(b) for b in br] for br in mod.branches]
        self.rows = []
        if mod.fuse_layers is not None:
            for i, row in enumerate(mod.fuse_layers):
                self.rows.append((i, [(j, _ChainT(row[j])) for j in range(i)],
                                  [(j, _ConvBN(row[j][0], row[j][1], False)) for j in range(i + 1, len(row))]))

    def forward(self, xs):
        xs = list(xs)
        for i, br in enumerate(self.branches):
            for blk in br:
                xs[i] = blk.forward(xs[i])
        if not self.rows:
            return xs
        out = []
        for i, downs, ups in self.rows:
            acc = xs[i]
            for j, chain in downs:
                acc = chain.forward(xs[j], skip=acc)
            out.append(_tap(vh.fuse_upsample_add(acc, [(cb.forward(xs[j]), j - i) for j, cb in ups], relu=True)))
        self.saved = out
        return out

    def backward(self, dys, grads):
        nb = len(self.branches)
        if self.rows:
            ys, self.saved = self.saved, None
            dxs = [None] * nb
            for i, downs, ups in self.rows:                # identity terms first: they seed the accumulators
                dxs[i] = vh.relu_bwd(dys[i], ys[i])
            gs = list(dxs)
            for i, downs, ups in self.rows:
                for j, chain in downs:
                    dxs[j] = chain.backward(gs[i], grads, dx_residual=dxs[j])
                for j, cb in ups:
                    dxs[j] = cb.backward(vh.upsample_nearest_bwd(dys[i], ys[i], j - i), grads, dx_residual=dxs[j])[0]
        else:
            dxs = list(dys)
        for i, br in enumerate(self.branches):
            dxs[i] = _chain_blocks_backward(br, dxs[i], grads, None)
        return dxs


class HRNetTrainer:
    """Tape-based forward/backward of PoseHighResolutionNet in training mode (hrnet.py:421-456)."""

    def __init__(self, m):
        self.stem1 = _ConvBN(m.conv1, m.bn1, True, need_dx=False)
        self.stem2 = _ConvBN(m.conv2, m.bn2, True)
        self.layer1 = [_block_t(b) for b in m.layer1]
        self.stages = []
        for s in (2, 3, 4):
            trans = [None if t is None else _ChainT(t) for t in getattr(m, f"transition{s - 1}")]
            self.stages.append((trans, [_HRModuleT(mod) for mod in getattr(m, f"stage{s}")]))
        self.head = m.final_layer

    @_planned("forward")
    def forward(self, x_nchw):
        x = self.stem2.forward(self.stem1.forward(vh.nchw_to_nhwc(x_nchw, 4)))
        for b in self.layer1:
            x = b.forward(x)
        ys = [x]
        self.widths = []
        for trans, mods in self.stages:
            self.widths.append(len(ys))
            xs = [ys[i] if t is None else t.forward(ys[-1]) for i, t in enumerate(trans)]
            for mod in mods:
                xs = mod.forward(xs)
            ys = xs
        self.head_in = ys[0]
        j, _, k, _ = self.head.weight.shape
        _, hb = vh.bn_fold(None, None, None, None, 0.0, self.head.bias.detach(), channels=j)
        _flush_batch_counters()
        return vh.conv2d_fwd(ys[0], vh.pack_conv_weight(self.head.weight.detach()), None, hb, j, k, k, 1, k // 2, False, out_nchw=True)

    @_planned("backward")
    def backward(self, dout_nchw, arena=None, overlap=False):
        grads = _Grads(arena, False)                  # the branches' gradients do not complete in arena order: one reduce at the end
        j, cin, k, _ = self.head.weight.shape
        dy = vh.nchw_to_nhwc(dout_nchw.contiguous(), 32)
        grads[self.head.bias] = vh.col_sum(dy)[:j].contiguous()
        grads[self.head.weight] = vh.conv2d_wgrad(self.head_in, dy, j, cin, k, k, 1, k // 2, out=_gout(grads, self.head.weight))
        wd = vh.pack_dgrad_weight(self.head.weight.detach(), _flipped_taps(k, k), cout_k=32)
        n, h, w, _ = self.head_in.shape
        dys = [vh.conv2d_fwd_ex(dy, wd, cin, k, k, 1, k // 2, k // 2, h, w, h, w, 1, 1, 0, 0)]
        self.head_in = None
        for (trans, mods), width in zip(reversed(self.stages), reversed(self.widths)):
            for mod in reversed(mods):
                dys = mod.backward(dys, grads)
            prev = [None] * width
            for i, t in enumerate(trans):                  # identity inputs first, then the paths that branch off ys[-1]
                if t is None:
                    prev[i] = dys[i]
            for i, t in enumerate(trans):
                if t is not None:
                    prev[width - 1] = t.backward(dys[i], grads, dx_residual=prev[width - 1])
            dys = prev
        dx = _chain_blocks_backward(self.layer1, dys[0], grads, None)
        dx, _ = self.stem2.backward(dx, grads)
        self.stem1.backward(dx, grads)
        _side.join()
        return grads.close()


class _TrainFn(torch.autograd.Function):
    """Bridges the HIP forward/backward into torch autograd so that the reference's
    `loss.backward()` fills `.grad` of every parameter (ActiveLearning.py:669-673)."""

    @staticmethod
    def forward(ctx, x, trainer, *params):
        ctx.trainer, ctx.params = trainer, params
        with torch.no_grad():
            return trainer.forward(x)

    @staticmethod
    def backward(ctx, dout):
        with torch.no_grad():
            grads = ctx.trainer.backward(dout.contiguous().float())
        return (None, None) + tuple(grads.get(p) for p in ctx.params)


def trainer_for(m: nn.Module):
    """The (cached) tape trainer of a pose network."""
    from .fastpose import FastPose
    from .hrnet import PoseHighResolutionNet
    from .simplepose import SimplePose
    tr = m.__dict__.get("_vatl_trainer")
    if tr is None:
        if isinstance(m, SimplePose):
            tr = SimplePoseTrainer(m)
        elif isinstance(m, FastPose):
            tr = FastPoseTrainer(m)
        elif isinstance(m, PoseHighResolutionNet):
            tr = HRNetTrainer(m)
        else:
            raise NotImplementedError(f"no training-mode HIP path for {type(m).__name__}")
        m.__dict__["_vatl_trainer"] = tr
    return tr


def arena_for(m: nn.Module):
    """The (cached) flat gradient arena of a pose network: ``p.grad`` of every trainable parameter is a slice of it."""
    from active_learning.distributed import GradArena
    params = [p for p in m.parameters() if p.requires_grad]
    ar = m.__dict__.get("_vatl_arena")
    if ar is None or len(ar.params) != len(params) or any(a is not b for a, b in zip(ar.params, params)) or ar.flat.device != params[0].device:
        ar = GradArena(params, device=params[0].device)
        m.__dict__["_vatl_arena"] = ar
    return ar


def forward_train(m: nn.Module, x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise vh.VatlError("the pose network trains on MI355X only (there is deliberately no CPU fallback)")
    tr = trainer_for(m)
    params = tuple(p for p in m.parameters() if p.requires_grad)
    return _TrainFn.apply(x.detach().float().contiguous(), tr, *params)
