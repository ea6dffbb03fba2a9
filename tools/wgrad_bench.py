#!/usr/bin/env python3
"""Weight-gradient kernel micro-benchmark on representative SimplePose-R50 shapes (B = 120 fine-tune batch).

    python tools/wgrad_bench.py [--batch 120] [--blocks 512,1024,2048]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

SHAPES = [  # name, H, W, Cin, Cout, k, stride, pad
    ("l1.c1 1x1 256->64 @64x48", 64, 48, 256, 64, 1, 1, 0),
    ("l1.c2 3x3 64->64 @64x48", 64, 48, 64, 64, 3, 1, 1),
    ("l1.c3 1x1 64->256 @64x48", 64, 48, 64, 256, 1, 1, 0),
    ("l2.c2 3x3 128->128 @32x24", 32, 24, 128, 128, 3, 1, 1),
    ("l3.c1 1x1 1024->256 @16x12", 16, 12, 1024, 256, 1, 1, 0),
    ("l3.c2 3x3 256->256 @16x12", 16, 12, 256, 256, 3, 1, 1),
    ("l3.c3 1x1 256->1024 @16x12", 16, 12, 256, 1024, 1, 1, 0),
    ("l4.c2 3x3 512->512 @8x6", 8, 6, 512, 512, 3, 1, 1),
    ("l4.c3 1x1 512->2048 @8x6", 8, 6, 512, 2048, 1, 1, 0),
    # FastPose-R152 at 384x288 (use --batch 32 --only r152)
    ("r152.l2.c2 3x3 128->128 @48x36", 48, 36, 128, 128, 3, 1, 1),
    ("r152.l3.c1 1x1 1024->256 @24x18", 24, 18, 1024, 256, 1, 1, 0),
    ("r152.l3.c2 3x3 256->256 @24x18", 24, 18, 256, 256, 3, 1, 1),
    ("r152.l3.c3 1x1 256->1024 @24x18", 24, 18, 256, 1024, 1, 1, 0),
    ("r152.l4.c2 3x3 512->512 @12x9", 12, 9, 512, 512, 3, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=120)
    ap.add_argument("--blocks", default="512,1024,2048")
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--ablate", type=int, default=0, help="vatl_tune_set(4, bits): 1 = no atomic epilogue (wrong results)")
    a = ap.parse_args()
    import vatl_hip as vh
    dev = torch.device("cuda:0")
    if a.ablate:
        vh.tune_set(4, a.ablate)
    for name, h, w, cin, cout, k, stride, pad in SHAPES:
        if a.only and a.only not in name:
            continue
        x = torch.randn((a.batch, h, w, cin), device=dev)
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        dz = torch.randn((a.batch, ho, wo, cout), device=dev)
        flop = 2.0 * a.batch * ho * wo * cout * cin * k * k
        row = {"shape": name, "ideal_us": round(flop / 157.3e12 * 1e6, 1)}
        for blocks in [int(b) for b in a.blocks.split(",")]:
            if blocks:
                vh.tune_set(3, blocks)
            for _ in range(3):
                vh.conv2d_wgrad(x, dz, cout, cin, k, k, stride, pad)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                vh.conv2d_wgrad(x, dz, cout, cin, k, k, stride, pad)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / a.iters
            row[f"us@{blocks}"] = round(us, 1)
            row[f"TF@{blocks}"] = round(flop / us / 1e6, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
