#!/usr/bin/env python3
"""Winograd weight gradients against the implicit-GEMM weight gradients on the fine-tune shapes: error against float64 (small batch) and time.

    python tools/wino_wgrad_bench.py [--batch 120] [--iters 5] [--blocks 1024,...]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import vatl_hip as vh  # noqa: E402
from wino_bench import DECONVS, SHAPES, timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=120)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", default="")
    ap.add_argument("--blocks", default="", help="comma list of target block counts to A/B (vatl_tune_set(19, v))")
    ap.add_argument("--halves", type=int, default=0, help="vatl_tune_set(23, v): gradient-channel halves per block of the Winograd weight-gradient kernel (1 or 2)")
    ap.add_argument("--table", default="", help="comma list of 0/1 to A/B: staging-address tables of the Winograd weight-gradient kernel (vatl_tune_set(25, v))")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.halves:
        vh.tune_set(23, a.halves)
    warm = torch.randn((4096, 4096), device=dev)
    for _ in range(100):
        warm @ warm
    torch.cuda.synchronize()
    g = torch.Generator(device="cpu").manual_seed(5)
    names = a.layers.split(",") if a.layers else list(SHAPES) + list(DECONVS)
    settings = [("blocks", int(v)) for v in a.blocks.split(",")] if a.blocks else ([("table", int(v)) for v in a.table.split(",")] * 2 if a.table else [("", 0)])
    for kind, blocks in settings:
        if kind == "blocks":
            vh.tune_set(19, blocks)
            print(f"--- target blocks {blocks}")
        elif kind == "table":
            vh.tune_set(25, blocks)
            print(f"--- staging-address tables {blocks}")
        for name in names:
            b = a.batch
            if name in DECONVS:
                h, w, cin, cout = DECONVS[name]
                x = torch.randn((b, h, w, cin), generator=g).to(dev)
                dy = torch.randn((b, 2 * h, 2 * w, cout), generator=g).to(dev)
                d1, d2 = vh.deconv4x4s2_wgrad(x, dy), vh.deconv4x4s2_winograd_wgrad(x, dy)
                td = timed(lambda: vh.deconv4x4s2_wgrad(x, dy), a.iters)
                tw = timed(lambda: vh.deconv4x4s2_winograd_wgrad(x, dy), a.iters)
                fl = 2.0 * b * h * w * 4 * cout * cin * 4
            else:
                h, w, cin, cout, _ = SHAPES[name]
                x = torch.randn((b, h, w, cin), generator=g).to(dev)
                dy = torch.randn((b, h, w, cout), generator=g).to(dev)
                d1, d2 = vh.conv2d_wgrad(x, dy, cout, cin, 3, 3, 1, 1), vh.conv3x3_winograd_wgrad(x, dy)
                td = timed(lambda: vh.conv2d_wgrad(x, dy, cout, cin, 3, 3, 1, 1), a.iters)
                tw = timed(lambda: vh.conv3x3_winograd_wgrad(x, dy), a.iters)
                fl = 2.0 * b * h * w * cout * cin * 9
            diff = (d1 - d2).abs().max().item() / d1.abs().max().item()
            print(f"{name:11s} B={b:4d} direct {td:8.1f} us {fl / td / 1e6:6.1f} TF/s | winograd {tw:8.1f} us {fl / tw / 1e6:6.1f} TF/s | rel diff {diff:.2e}  speed-up {td / tw:.2f}x", flush=True)


if __name__ == "__main__":
    main()
