#!/usr/bin/env python3
"""Where do the transposed convs' Winograd launches spend their time?  Profiling-library ablations like tools/ablate_wino.py (WRONG results by construction):

    VATL_HIP_LIB=vatl4pose-wacv2024_amd/vatl_hip/libvatl_hip_ablation.so VATL_ALLOW_ABLATION=1 python tools/ablate_deconv.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import vatl_hip as vh
from wino_bench import DECONVS, timed
CASES = [(0, "as is"), (1, "no output transform"), (2, "no LDS reads"), (4, "no filter loads"), (8, "no staging DMA"), (16, "no barriers"),
         (24, "no DMA, no barriers"), (2 | 4 | 8 | 16, "MFMAs + output transform"), (31, "MFMAs only")]
dev = torch.device("cuda:0")
warm = torch.randn((4096, 4096), device=dev)
for _ in range(100): warm @ warm
b = 1024
for name in ("deconv3", "deconv2", "deconv1"):
    h, w, cin, cout = DECONVS[name]
    x = torch.randn((b, h, w, cin), device=dev)
    wt = torch.randn((cin, cout, 4, 4), device=dev) * 0.03
    up = vh.pack_winograd_deconv_weight(wt)
    sc = torch.rand(cout, device=dev) + 0.5; bi = torch.randn(cout, device=dev)
    floor = 2.0 * b * h * w * cout * cin * 4 * 4 / 2.25 / 157.3e12 * 1e6
    row = []
    for bits, label in CASES:
        vh.tune_set(17, bits)
        row.append(f"{label}: {timed(lambda: vh.deconv4x4s2_winograd_fwd(x, up, sc, bi, cout, True), 3):.0f}")
    vh.tune_set(17, 0)
    print(f"{name} (MFMA floor {floor:.0f} us)  " + " | ".join(row), flush=True)
