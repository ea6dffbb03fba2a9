#!/usr/bin/env python3
"""Where does a short-K 1x1 layer spend its time?  Profiling-library ablations of the generic 128x128 kernel (WRONG results by
construction): no epilogue, one k-tile only, no global loads / LDS writes, no barrier, no fragment reads.
    VATL_HIP_LIB=vatl4pose-wacv2024_amd/vatl_hip/libvatl_hip_ablation.so VATL_ALLOW_ABLATION=1 python tools/ablate_1x1.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)
import torch
import vatl_hip as vh
dev = torch.device("cuda:0")
shapes = {"l1.c3nr": (64, 48, 64, 256), "l2.c3nr": (32, 24, 128, 512), "l3.c3nr": (16, 12, 256, 1024), "l4.c1": (8, 6, 2048, 512)}
B = 1024
w_ = torch.randn((4096, 4096), device=dev)
for _ in range(100): w_ @ w_
torch.cuda.synchronize()
def timeit(fn, it=10):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for name, (H, W, cin, cout) in shapes.items():
    x = torch.randn((B, H, W, cin), device=dev)
    w = vh.pack_conv_weight(torch.randn((cout, cin, 1, 1), device=dev) * 0.01)
    sc = torch.ones(cout, device=dev); bi = torch.zeros(cout, device=dev)
    fn = lambda: vh.conv2d_fwd(x, w, sc, bi, cout, 1, 1, 1, 0, True)
    flops = 2.0 * B * H * W * cin * cout
    out = []
    vh.tune_set(7, 1); vh.tune_set(0, 4); vh.tune_set(6, 0)
    out.append(("persistent", timeit(fn)))
    for abl, tag in ((1, "P no epilogue"), (4, "P stores dropped"), (8, "P loads dropped"), (12, "P no HBM traffic"), (5, "P no epilogue, no stores"), (9, "P no epilogue, no loads"), (2, "P two k-tiles")):
        vh.tune_set(6, abl)
        out.append((tag, timeit(fn)))
    vh.tune_set(6, 0)
    vh.tune_set(7, 0)
    for var, abl, tag in ((4, 0, "generic VAR4"), (2, 0, "VAR2"), (4, 1, "VAR4 no epilogue"), (4, 2, "VAR4 one k-tile"), (4, 3, "one k-tile, no epilogue"),
                          (10, 0, "no global loads"), (10, 1, "no global loads, no epilogue"), (11, 0, "no barrier"), (12, 0, "no fragment reads"), (13, 0, "loads kept, no LDS writes")):
        vh.tune_set(0, var); vh.tune_set(6, abl)
        out.append((tag, timeit(fn)))
    vh.tune_set(0, 4); vh.tune_set(6, 0); vh.tune_set(7, 1)
    print(name, f"K={cin} N={cout}", " | ".join(f"{t}: {us:.0f} us ({flops / us / 1e6:.0f} TF/s)" for t, us in out), flush=True)
