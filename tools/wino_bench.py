#!/usr/bin/env python3
"""Winograd F(2x2,3x3) route against the implicit GEMM on the 3x3 / stride-1 layer shapes: error against float64 and time.

    python tools/wino_bench.py [--batch 1024] [--iters 5] [--layers l1.c2,...] [--check 8]

For every shape: max |error| of both routes against a float64 convolution (on `--check` crops), then `iters` launches of
each route between HIP events (TFLOP/s are algorithmic = direct-convolution FLOPs for both).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import vatl_hip as vh  # noqa: E402

SHAPES = {  # name: (H, W, Cin, Cout, residual)
    "l1.c2": (64, 48, 64, 64, False), "l2.c2": (32, 24, 128, 128, False), "l3.c2": (16, 12, 256, 256, False), "l4.c2": (8, 6, 512, 512, False),
    "hr.b32": (64, 48, 32, 32, True), "hr.b64": (32, 24, 64, 64, True), "hr.b128": (16, 12, 128, 128, True), "hr.b256": (8, 6, 256, 256, True),
    "r152.l2.c2": (48, 36, 128, 128, False), "r152.l3.c2": (24, 18, 256, 256, False), "r152.l4.c2": (12, 9, 512, 512, False),
}
DECONVS = {"deconv1": (8, 6, 2048, 256), "deconv2": (16, 12, 256, 256), "deconv3": (32, 24, 256, 256)}


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", default="")
    ap.add_argument("--check", type=int, default=8)
    ap.add_argument("--group-kb", type=int, default=-1, help="vatl_tune_set(18, v): KB of filter slices per group of the Winograd tile order")
    ap.add_argument("--halves", type=int, default=0, help="vatl_tune_set(21, v): 32-channel filter halves per Winograd block (1, or 2 where the layer allows)")
    ap.add_argument("--persist", type=int, default=-1, help="vatl_tune_set(22, v): layers with at most v 16-channel stages take the persistent Winograd route (0 = never)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.persist >= 0:
        vh.tune_set(22, a.persist)
    if a.group_kb >= 0:
        vh.tune_set(18, a.group_kb)
    if a.halves:
        vh.tune_set(21, a.halves)
    warm = torch.randn((4096, 4096), device=dev)
    for _ in range(100):
        warm @ warm
    torch.cuda.synchronize()
    g = torch.Generator(device="cpu").manual_seed(5)
    names = a.layers.split(",") if a.layers else list(SHAPES) + list(DECONVS)
    for name in names:
        if name in DECONVS:
            h, w, cin, cout = DECONVS[name]
            b = a.batch
            x = torch.randn((b, h, w, cin), generator=g).to(dev)
            wt = (torch.randn((cin, cout, 4, 4), generator=g) * (2.0 / (4 * cin)) ** 0.5).to(dev)
            sc = (torch.rand(cout, generator=g) + 0.5).to(dev)
            bi = torch.randn(cout, generator=g).to(dev)
            wp, up = vh.pack_deconv_weight(wt), vh.pack_winograd_deconv_weight(wt)
            yd = vh.deconv4x4s2_fwd(x, wp, sc, bi, cout, True)
            yw = vh.deconv4x4s2_winograd_fwd(x, up, sc, bi, cout, True)
            k = min(a.check, b)
            ref = torch.nn.functional.conv_transpose2d(x[:k].permute(0, 3, 1, 2).double(), wt.double(), None, 2, 1) * sc.double().view(1, -1, 1, 1) + bi.double().view(1, -1, 1, 1)
            ref = ref.clamp_min(0).permute(0, 2, 3, 1)
            ed, ew = (yd[:k].double() - ref).abs().max().item(), (yw[:k].double() - ref).abs().max().item()
            td = timed(lambda: vh.deconv4x4s2_fwd(x, wp, sc, bi, cout, True), a.iters)
            tw = timed(lambda: vh.deconv4x4s2_winograd_fwd(x, up, sc, bi, cout, True), a.iters)
            fl = 2.0 * b * h * w * 4 * cout * cin * 4
            print(f"{name:11s} B={b:5d} direct {td:8.1f} us {fl / td / 1e6:6.1f} TF/s err {ed:.2e} | winograd {tw:8.1f} us {fl / tw / 1e6:6.1f} TF/s err {ew:.2e} "
                  f"| ref max {ref.abs().max().item():.2f}  speed-up {td / tw:.2f}x", flush=True)
            continue
        h, w, cin, cout, res = SHAPES[name]
        b = a.batch
        x = torch.randn((b, h, w, cin), generator=g).to(dev)
        wt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
        sc = (torch.rand(cout, generator=g) + 0.5).to(dev)
        bi = torch.randn(cout, generator=g).to(dev)
        r = torch.randn((b, h, w, cout), generator=g).to(dev) if res else None
        wp = vh.pack_conv_weight(wt)
        up = vh.pack_winograd_weight(wt)
        yd = vh.conv2d_fwd(x, wp, sc, bi, cout, 3, 3, 1, 1, True, residual=r)
        yw = vh.conv3x3_winograd_fwd(x, up, sc, bi, cout, True, residual=r)
        k = min(a.check, b)
        ref = torch.nn.functional.conv2d(x[:k].permute(0, 3, 1, 2).double(), wt.double(), padding=1) * sc.double().view(1, -1, 1, 1) + bi.double().view(1, -1, 1, 1)
        if res:
            ref = ref + r[:k].permute(0, 3, 1, 2).double()
        ref = ref.clamp_min(0).permute(0, 2, 3, 1)
        ed = (yd[:k].double() - ref).abs().max().item()
        ew = (yw[:k].double() - ref).abs().max().item()
        full = (yd - yw).abs().max().item()
        td = timed(lambda: vh.conv2d_fwd(x, wp, sc, bi, cout, 3, 3, 1, 1, True, residual=r), a.iters)
        tw = timed(lambda: vh.conv3x3_winograd_fwd(x, up, sc, bi, cout, True, residual=r), a.iters)
        fl = 2.0 * b * h * w * cout * cin * 9
        print(f"{name:11s} B={b:5d} direct {td:8.1f} us {fl / td / 1e6:6.1f} TF/s err {ed:.2e} | winograd {tw:8.1f} us {fl / tw / 1e6:6.1f} TF/s err {ew:.2e} "
              f"| max |direct - winograd| {full:.2e}  ref max {ref.abs().max().item():.2f}  speed-up {td / tw:.2f}x", flush=True)


if __name__ == "__main__":
    main()
