#!/usr/bin/env python3
"""Is the HRNet branch conv (Winograd, 32 / 64 channels) bound by HBM?  Per-launch time of the same layer with tensors that stream
from HBM (1024 crops), that fit the Infinity Cache (128 crops) and that fit L2 (16 crops), with and without the residual read,
and of the BasicBlock pair conv1 -> conv2(+x) back to back.  Prints us per 1024-crop equivalent."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vatl_hip as vh  # noqa: E402


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    warm = torch.randn((4096, 4096), device=dev)
    for _ in range(100):
        warm @ warm
    g = torch.Generator(device="cpu").manual_seed(5)
    for name, (h, w, c) in (("hr.b32", (64, 48, 32)), ("hr.b64", (32, 24, 64)), ("hr.b128", (16, 12, 128))):
        wt = (torch.randn((c, c, 3, 3), generator=g) * (2.0 / (9 * c)) ** 0.5).to(dev)
        u = vh.pack_winograd_weight(wt)
        sc = (torch.rand(c, generator=g) + 0.5).to(dev); bi = torch.randn(c, generator=g).to(dev)
        for b in (1024, 256, 128, 32):
            x = torch.randn((b, h, w, c), device=dev)
            r = torch.randn((b, h, w, c), device=dev)
            mid = torch.empty_like(x); y = torch.empty_like(x)
            it = max(5, 4096 // b)
            t0 = timed(lambda: vh.conv3x3_winograd_fwd(x, u, sc, bi, c, True, out=y), it)
            t1 = timed(lambda: vh.conv3x3_winograd_fwd(x, u, sc, bi, c, True, residual=r, out=y), it)
            t2 = timed(lambda: vh.conv3x3_winograd_fwd(x, u, sc, bi, c, True, residual=x, out=y), it)

            def pair():
                vh.conv3x3_winograd_fwd(x, u, sc, bi, c, True, out=mid)
                vh.conv3x3_winograd_fwd(mid, u, sc, bi, c, True, residual=x, out=y)
            t3 = timed(pair, it)
            k = 1024 / b
            print(f"{name} B={b:5d} tensor {x.numel() * 4 / 2**20:7.1f} MB | no-res {t0 * k:7.1f}  res {t1 * k:7.1f}  res=x {t2 * k:7.1f}  pair {t3 * k:7.1f}  (us per 1024 crops; one launch = {t1:.1f} us)", flush=True)


if __name__ == "__main__":
    main()
