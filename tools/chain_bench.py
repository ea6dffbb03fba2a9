#!/usr/bin/env python3
"""conv3 (+ skip + ReLU) of one bottleneck chained with conv1 of the next (vatl_bottleneck_chain_fwd, csrc/bottleneck_chain.hip) against the two tiled launches.
usage: chain_bench.py [crops of 64x48 pixels, default 1024]"""
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
    warm = torch.randn((4096, 4096), device=dev)
    for _ in range(100):
        warm @ warm
    n, h, w = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 64, 48
    g = torch.Generator(device=dev); g.manual_seed(3)
    a = torch.randn((n, h, w, 64), device=dev, generator=g)
    res = torch.randn((n, h, w, 256), device=dev, generator=g)
    w3 = torch.randn((256, 64, 1, 1), device=dev, generator=g) / 8
    w1 = torch.randn((64, 256, 1, 1), device=dev, generator=g) / 16
    s3 = torch.rand(256, device=dev, generator=g) + 0.5; b3 = torch.randn(256, device=dev, generator=g)
    s1 = torch.rand(64, device=dev, generator=g) + 0.5; b1 = torch.randn(64, device=dev, generator=g)
    w3p, w1p = vh.pack_conv_weight(w3), vh.pack_conv_weight(w1)
    t_ref = vh.conv2d_fwd(a, w3p, s3, b3, 256, 1, 1, 1, 0, True, residual=res)
    o_ref = vh.conv2d_fwd(t_ref, w1p, s1, b1, 64, 1, 1, 1, 0, True)
    t = torch.empty_like(t_ref); o = torch.empty_like(o_ref)

    def fused(second=True):
        if second:
            vh.bottleneck_chain_fwd(a, w3p, s3, b3, res, w1p, s1, b1, out=t, y1_out=o)
        else:
            vh.bottleneck_chain_fwd(a, w3p, s3, b3, res, out=t)
    fused(); torch.cuda.synchronize()
    print("T bit-identical:", bool(torch.equal(t, t_ref)), " out2 rel err:", float((o - o_ref).abs().max() / o_ref.abs().max()))
    t.zero_(); fused(False); torch.cuda.synchronize()
    print("first GEMM only, T bit-identical:", bool(torch.equal(t, t_ref)))

    def two():
        vh.conv2d_fwd(a, w3p, s3, b3, 256, 1, 1, 1, 0, True, residual=res, out=t)
        vh.conv2d_fwd(t, w1p, s1, b1, 64, 1, 1, 1, 0, True, out=o)
    for rep in range(2):
        t2 = timed(two); tc3 = timed(lambda: vh.conv2d_fwd(a, w3p, s3, b3, 256, 1, 1, 1, 0, True, residual=res, out=t))
        tf = timed(fused); tf1 = timed(lambda: fused(False))
        print(f"B={n}: tiled conv3 {tc3:7.1f} us, conv3 + conv1 {t2:7.1f} us | fused first GEMM only {tf1:7.1f} us, fused pair {tf:7.1f} us  ({t2 / tf:.2f}x)", flush=True)


if __name__ == "__main__":
    main()
