#!/usr/bin/env python3
"""The row-streaming K = 128 GEMM (vatl_conv1x1_rows_fwd, csrc/conv1x1_rows.hip) against the tiled kernels on the layers it serves.
usage: rows_bench.py [crops, default 1024]"""
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
    g = torch.Generator(device=dev); g.manual_seed(3)
    for name, (h, w, cout, res, kin) in {"l2.n.c3 (128 -> 512 + skip)": (32, 24, 512, True, 128), "se.l2.c3 (128 -> 512)": (32, 24, 512, False, 128), "r152@384 l2.n.c3": (48, 36, 512, True, 128),
                                        "l3.n.c3 (256 -> 1024 + skip)": (16, 12, 1024, True, 256), "se.l3.c3 (256 -> 1024)": (16, 12, 1024, False, 256),
                                        "r152@384 l3.n.c3": (24, 18, 1024, True, 256), "hr.layer1-like (256 -> 256)": (64, 48, 256, False, 256)}.items():
        nn = n if h in (32, 16) else max(n // 4, 1)
        a = torch.randn((nn, h, w, kin), device=dev, generator=g)
        r = torch.randn((nn, h, w, cout), device=dev, generator=g) if res else None
        wt = vh.pack_conv_weight(torch.randn((cout, kin, 1, 1), device=dev, generator=g) / 11)
        sc = torch.rand(cout, device=dev, generator=g) + 0.5; bi = torch.randn(cout, device=dev, generator=g)
        y = torch.empty((nn, h, w, cout), device=dev)
        for rep in range(2):
            t0 = timed(lambda: vh.conv2d_fwd(a, wt, sc, bi, cout, 1, 1, 1, 0, True, residual=r, out=y))
            t1 = timed(lambda: vh.conv1x1_rows_fwd(a, wt, sc, bi, cout, True, residual=r, out=y))
            fl = 2.0 * nn * h * w * kin * cout
            print(f"{name:30s} B={nn}: tiled {t0:7.1f} us ({fl / t0 / 1e6:5.1f} TF/s)  rows {t1:7.1f} us ({fl / t1 / 1e6:5.1f} TF/s)  {t0 / t1:.2f}x", flush=True)
    h, w, cout = 64, 48, 256
    a, x = torch.randn((n, h, w, 64), device=dev, generator=g), torch.randn((n, h, w, 64), device=dev, generator=g)
    w1, w2 = torch.randn((cout, 64, 1, 1), device=dev, generator=g) / 8, torch.randn((cout, 64, 1, 1), device=dev, generator=g) / 8
    s1, b1, s2, b2 = (torch.rand(cout, device=dev, generator=g) + 0.5 for _ in range(4))
    wp, bias = vh.pack_conv1x1_dual_weight(w1, s1, b1, w2, s2, b2)
    y = torch.empty((n, h, w, cout), device=dev)
    for rep in range(2):
        t0 = timed(lambda: vh.conv1x1_dual_fwd(a, x, wp, bias, cout, 1, True, out=y))
        t1 = timed(lambda: vh.conv1x1_rows_fwd(a, wp, None, bias, cout, True, x2=x, out=y))
        fl = 2.0 * n * h * w * 128 * cout
        print(f"{'l1.0.c3+p (64 + 64 -> 256)':30s} B={n}: tiled {t0:7.1f} us ({fl / t0 / 1e6:5.1f} TF/s)  rows {t1:7.1f} us ({fl / t1 / 1e6:5.1f} TF/s)  {t0 / t1:.2f}x", flush=True)


if __name__ == "__main__":
    main()
