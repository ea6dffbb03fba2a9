#!/usr/bin/env python3
"""HBM roofline of the training-mode BatchNorm kernels on the SimplePose-R50 layer shapes.

    python tools/bn_bench.py [--batch 120] [--iters 10]
Per shape (rows M = batch * H * W, channels C): the backward pair `vatl_bn_train_bwd_relu` (column reduction over dy and z,
finalize, apply = 2 + 3 tensor passes) and the forward apply `vatl_scale_bias_act` (2 passes), in us and GB/s.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

SHAPES = [("stem", 128 * 96, 64), ("l1.c1/c2", 64 * 48, 64), ("l1.c3", 64 * 48, 256), ("l2.c1@64x48", 64 * 48, 128), ("l2.c2", 32 * 24, 128),
          ("l2.c3", 32 * 24, 512), ("l3.c1@32x24", 32 * 24, 256), ("l3.c2", 16 * 12, 256), ("l3.c3", 16 * 12, 1024), ("l4.c1@16x12", 16 * 12, 512),
          ("l4.c2", 8 * 6, 512), ("l4.c3", 8 * 6, 2048), ("deconv1", 16 * 12, 256), ("deconv2", 32 * 24, 256), ("deconv3", 64 * 48, 256)]


def timed(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=120)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    import vatl_hip as vh
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(3)
    spill = torch.empty(96 << 20, device=dev)                     # 384 MB written between launches: nothing stays in the 256 MB Infinity Cache
    tot = [0.0, 0.0, 0.0]
    for name, hw, c in SHAPES:
        m = a.batch * hw
        z = torch.randn((m, c), device=dev, generator=g)
        dy = torch.randn((m, c), device=dev, generator=g)
        gamma = torch.rand(c, device=dev, generator=g) + 0.5
        mean, invstd = z.mean(0), 1.0 / z.std(0)
        scale, bias = gamma * invstd, -mean * gamma * invstd

        def bwd():
            spill.zero_()
            return vh.bn_train_bwd_relu(dy, scale, bias, z, gamma, mean, invstd)

        def fwd():
            spill.zero_()
            return vh.scale_bias_act(z, scale, bias)
        base = timed(lambda: spill.zero_(), a.iters)
        tb, tf = timed(bwd, a.iters) - base, timed(fwd, a.iters) - base
        byts = m * c * 4
        tot[0] += tb; tot[1] += tf; tot[2] += byts
        print(json.dumps({"layer": name, "M": m, "C": c, "bwd_us": round(tb, 1), "bwd_GBps": round(5 * byts / tb / 1e3, 1),
                          "fwd_apply_us": round(tf, 1), "fwd_GBps": round(2 * byts / tf / 1e3, 1)}))
    print(json.dumps({"sum_bwd_us": round(tot[0], 1), "sum_fwd_us": round(tot[1], 1), "bwd_GBps": round(5 * tot[2] / tot[0] / 1e3, 1)}))


if __name__ == "__main__":
    main()
