#!/usr/bin/env python3
"""Winograd F(4x4,3x3) (csrc/winograd_f4.hip) against the F(2x2,3x3) route on the layers it serves: error of both against float64, time of both.

    python tools/f4_bench.py [--batch 1024] [--iters 10]
"""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)
import torch
import vatl_hip as vh

SHAPES = {"l3.c2": (16, 12, 256, 256, False), "l2.c2": (32, 24, 128, 128, False), "l1.c2": (64, 48, 64, 64, False), "hr.b128": (16, 12, 128, 128, True),
          "hr.b64": (32, 24, 64, 64, True), "r152.l2.c2": (48, 36, 128, 128, False), "duc1": (24, 18, 512, 1024, False)}
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--layers", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
warm = torch.randn((4096, 4096), device=dev)
for _ in range(100):
    warm @ warm
torch.cuda.synchronize()


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g = torch.Generator(device="cpu").manual_seed(5)
for name in (a.layers.split(",") if a.layers else SHAPES):
    h, w, cin, cout, skip = SHAPES[name]
    b = a.batch if name != "duc1" else min(a.batch, 64)
    if not vh.conv3x3_winograd_f4_supported(b, h, w, cin, cout):
        print(f"{name}: not served"); continue
    x = torch.randn((b, h, w, cin), generator=g).to(dev)
    wt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
    sc, bi = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    res = torch.randn((b, h, w, cout), generator=g).to(dev) if skip else None
    u4, u2 = vh.pack_winograd_f4_weight(wt), vh.pack_winograd_weight(wt)
    y4 = vh.conv3x3_winograd_f4_fwd(x, u4, sc, bi, cout, True, residual=res)
    y2 = vh.conv3x3_winograd_fwd(x, u2, sc, bi, cout, True, residual=res)
    if cin == 32 and cout == 32:                           # the wave-private F(2x2) kernel is what these layers run on today
        u32 = vh.pack_winograd_c32_weight(wt)
        t32 = timed(lambda: vh.conv3x3_winograd_c32_fwd(x, u32, sc, bi, True, residual=res), a.iters)
        print(f"{name:11s} winograd_c32 {t32:8.1f} us", flush=True)
    k = min(4, b)
    ref = torch.nn.functional.conv2d(x[:k].permute(0, 3, 1, 2).double(), wt.double(), padding=1).permute(0, 2, 3, 1) * sc.double() + bi.double()
    if skip:
        ref = ref + res[:k].double()
    ref = ref.clamp_min(0)
    e4, e2 = ((y4[:k].double() - ref).abs().max() / ref.abs().max()).item(), ((y2[:k].double() - ref).abs().max() / ref.abs().max()).item()
    t4 = timed(lambda: vh.conv3x3_winograd_f4_fwd(x, u4, sc, bi, cout, True, residual=res), a.iters)
    t2 = timed(lambda: vh.conv3x3_winograd_fwd(x, u2, sc, bi, cout, True, residual=res), a.iters)
    fl = 2.0 * b * h * w * cout * cin * 9
    print(f"{name:11s} B={b:5d}  F(2x2) {t2:8.1f} us {fl / t2 / 1e6:6.1f} TF/s err {e2:.2e} | F(4x4) {t4:8.1f} us {fl / t4 / 1e6:6.1f} TF/s err {e4:.2e} | executed pipe share "
          f"{fl / 4 / t4 / 1e6 / 157.3:.2f} | speed-up {t2 / t4:.2f}x | bits {hashlib.sha1(y4.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
