#!/usr/bin/env python3
"""The wave-private Winograd kernel for 64 -> 64 channel 3x3 layers (vatl_conv3x3_winograd_c64_fwd, csrc/winograd_c64.hip) against the general Winograd route.
usage: c64_bench.py [crops, default 1024]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vatl_hip as vh  # noqa: E402


def timed(fn, it=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    warm = torch.randn((4096, 4096), device=dev)
    for _ in range(100):
        warm @ warm
    g = torch.Generator(device=dev); g.manual_seed(7)
    w = torch.randn((64, 64, 3, 3), device=dev, generator=g) * (2.0 / 576) ** 0.5
    sc = torch.rand(64, device=dev, generator=g) + 0.5; bi = torch.randn(64, device=dev, generator=g)
    u_old, u_new = vh.pack_winograd_weight(w), vh.pack_winograd_c64_weight(w)
    # small case against float64
    for (b, h, wd) in ((2, 8, 6), (1, 64, 48), (3, 10, 22)):
        x = torch.randn((b, h, wd, 64), device=dev, generator=g); r = torch.randn((b, h, wd, 64), device=dev, generator=g)
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().cpu(), padding=1).permute(0, 2, 3, 1)
        ref = torch.relu(ref * sc.double().cpu() + bi.double().cpu() + r.double().cpu())
        got = vh.conv3x3_winograd_c64_fwd(x, u_new, sc, bi, True, residual=r).double().cpu()
        old = vh.conv3x3_winograd_fwd(x, u_old, sc, bi, 64, True, residual=r).double().cpu()
        print(f"{b}x{h}x{wd}: new vs float64 {float((got - ref).abs().max() / ref.abs().max()):.2e}  general route vs float64 {float((old - ref).abs().max() / ref.abs().max()):.2e}", flush=True)
    for (hh, ww, tag) in ((32, 24, "hr.b64"), (64, 48, "l1.c2 ")):
        x = torch.randn((n, hh, ww, 64), device=dev, generator=g); r = torch.randn((n, hh, ww, 64), device=dev, generator=g)
        y = torch.empty_like(x)
        for res in (r, None):
            for rep in range(2):
                t0 = timed(lambda: vh.conv3x3_winograd_fwd(x, u_old, sc, bi, 64, True, residual=res, out=y))
                t1 = timed(lambda: vh.conv3x3_winograd_c64_fwd(x, u_new, sc, bi, True, residual=res, out=y))
                fl = 2.0 * n * hh * ww * 64 * 64 * 9
                print(f"{tag} {'+res' if res is not None else '    '} B={n}: general {t0:7.1f} us ({fl / t0 / 1e6:5.1f} TF/s)  wave-private {t1:7.1f} us ({fl / t1 / 1e6:5.1f} TF/s)  {t0 / t1:.2f}x", flush=True)

if __name__ == "__main__":
    main()
