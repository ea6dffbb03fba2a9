#!/usr/bin/env python3
"""The data-gradient launches with the BatchNorm-backward epilogue (vatl_conv2d_fwd_ex_bnbwd: three operand tensors + one store per
output tile next to a short-K GEMM) of the B = 120 fine-tune step, one shape at a time between HIP events.

    python tools/bnbwd_bench.py [--iters 20] [--bm 0|64|128] [--stagger PCT]      (--stagger needs the ablation library: VATL_HIP_LIB)

Prints us per launch, the direct-sum TFLOP/s and the HBM rate of the launch's compulsory traffic (x, residual, z, mask, store).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vatl_hip as vh  # noqa: E402

# (N, H, W, Cin, Cout, residual, mask_y): the 1x1 data gradients of configs[2] (Cin = the layer's output channels)
SHAPES = [(120, 16, 12, 256, 1024, True, True), (120, 16, 12, 1024, 256, False, False), (120, 64, 48, 64, 256, True, True),
          (120, 32, 24, 128, 512, True, True), (120, 32, 24, 512, 128, False, False), (120, 8, 6, 2048, 512, False, False),
          (120, 8, 6, 512, 2048, True, True), (120, 64, 48, 256, 64, False, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bm", type=int, default=0)
    ap.add_argument("--stagger", type=int, default=0)
    ap.add_argument("--ablate", type=int, default=0, help="vatl_tune_set(6, bits) of the ablation library: 1 = no epilogue, 2 = one k-tile only (wrong results)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.bm:
        vh.tune_set(5, a.bm)
    if a.stagger:
        vh.tune_set(2, a.stagger)
    if a.ablate:
        vh.tune_set(6, a.ablate)
    g = torch.Generator(device=dev); g.manual_seed(3)
    for n, h, w, cin, cout, has_res, has_mask in SHAPES:
        x = torch.randn((n, h, w, cin), device=dev, generator=g)
        wt = torch.randn((cout, cin, 1, 1), device=dev, generator=g) * 0.05
        wp = vh.pack_conv_weight(wt)
        z = torch.randn((n, h, w, cout), device=dev, generator=g)
        res = torch.randn_like(z) if has_res else None
        my = torch.randn_like(z) if has_mask else None
        mean, invstd = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
        sc, bi = (None, None) if has_mask else (torch.ones(cout, device=dev), torch.zeros(cout, device=dev))
        out = torch.empty_like(z)

        def run():
            spec = vh.BnBwdSpec(z, mean, invstd, mask_y=my, scale=sc, bias=bi)
            vh.conv2d_fwd_ex_bnbwd(x, wp, cout, 1, 1, 1, 0, 0, h, w, h, w, 1, 1, 0, 0, spec, out=out, residual=res)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        m = n * h * w
        flops = 2.0 * m * cin * cout
        traffic = 4.0 * m * (cin + cout * (2 + has_res + has_mask))
        print("bm=%-3d stagger=%-3d abl=%d  %4dx%2dx%2d %4d -> %4d res=%d mask=%d  %7.1f us  %6.1f TF/s  %5.2f TB/s" %
              (a.bm, a.stagger, a.ablate, n, h, w, cin, cout, has_res, has_mask, us, flops / us / 1e6, traffic / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
