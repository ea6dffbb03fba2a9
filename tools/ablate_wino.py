#!/usr/bin/env python3
"""Where does the Winograd kernel spend its time?  Profiling-library ablations (WRONG results by construction):

    VATL_HIP_LIB=vatl4pose-wacv2024_amd/vatl_hip/libvatl_hip_ablation.so VATL_ALLOW_ABLATION=1 python tools/ablate_wino.py

bits of vatl_tune_set(17, .): 1 no output transform, 2 no LDS reads / input transform, 4 no filter loads, 8 no staging DMA, 16 no barriers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import vatl_hip as vh  # noqa: E402
from wino_bench import SHAPES, timed  # noqa: E402

CASES = [(0, "as is"), (1, "no output transform"), (2, "no LDS reads"), (4, "no filter loads"), (8, "no staging DMA"), (16, "no barriers"),
         (24, "no DMA, no barriers"), (2 | 4 | 8 | 16, "MFMAs + output transform"), (31, "MFMAs only")]


def main():
    dev = torch.device("cuda:0")
    warm = torch.randn((4096, 4096), device=dev)
    for _ in range(100):
        warm @ warm
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["l1.c2", "l3.c2", "l4.c2", "hr.b32"]
    b = 1024
    for name in names:
        h, w, cin, cout, res = SHAPES[name]
        x = torch.randn((b, h, w, cin), device=dev)
        wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
        up = vh.pack_winograd_weight(wt)
        y = torch.empty((b, h, w, cout), device=dev)
        floor = 2.0 * b * h * w * cout * cin * 4 / 157.3e12 * 1e6
        row = []
        for bits, label in CASES:
            vh.tune_set(17, bits)
            row.append(f"{label}: {timed(lambda: vh.conv3x3_winograd_fwd(x, up, None, None, cout, True, out=y), 5):.0f}")
        vh.tune_set(17, 0)
        print(f"{name} (MFMA floor {floor:.0f} us)  " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
