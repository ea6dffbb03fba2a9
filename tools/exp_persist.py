#!/usr/bin/env python3
"""A/B of the persistent Winograd route (vatl_tune_set(22, v)) on the short-block 3x3 layers, same process, alternating."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vatl_hip as vh  # noqa: E402

SHAPES = {"hr.b32": (64, 48, 32, 32, True), "hr.b64": (32, 24, 64, 64, True), "hr.b128": (16, 12, 128, 128, True), "l1.c2": (64, 48, 64, 64, False),
          "l2.c2": (32, 24, 128, 128, False), "r152.l2.c2": (48, 36, 128, 128, False), "hr.t32": (64, 48, 256, 32, False)}


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
    batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "1024,120").split(",")]
    warm = torch.randn((4096, 4096), device=dev)
    for _ in range(100):
        warm @ warm
    g = torch.Generator(device="cpu").manual_seed(5)
    lib = vh.lib()
    for name, (h, w, cin, cout, res) in SHAPES.items():
        wt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
        u = vh.pack_winograd_weight(wt)
        sc = (torch.rand(cout, generator=g) + 0.5).to(dev); bi = torch.randn(cout, generator=g).to(dev)
        for b in batches:
            if name.startswith("r152") and b > 256:
                b = 256
            x = torch.randn((b, h, w, cin), device=dev)
            r = torch.randn((b, h, w, cout), device=dev) if res else None
            y = torch.empty((b, h, w, cout), device=dev)
            f = lambda: vh.conv3x3_winograd_fwd(x, u, sc, bi, cout, True, residual=r, out=y)
            ts = {}
            for rep in range(2):
                for v, pf in ((0, 0), (128, 0), (128, 3)):
                    vh.tune_set(22, v); vh.tune_set(24, pf)
                    t = timed(f, 10)
                    ts.setdefault((v, pf), []).append(t)
                    if v:
                        route = lib.vatl_winograd_last_route()
            vh.tune_set(22, 8); vh.tune_set(24, 2)
            a, c, d = min(ts[(0, 0)]), min(ts[(128, 0)]), min(ts[(128, 3)])
            print(f"{name:11s} B={b:5d} plain {a:8.1f} us  persistent {c:8.1f} us ({a / c:4.2f}x)  persistent + prefetch {d:8.1f} us ({a / d:4.2f}x)  route {route}", flush=True)


if __name__ == "__main__":
    main()
