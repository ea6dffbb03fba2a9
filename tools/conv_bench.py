#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernel on the SimplePose-R50 layer shapes.

    python tools/conv_bench.py [--batch 1024] [--iters 5] [--layers l3.n.c2,deconv3,...]

Each shape is launched `iters` times back to back between two HIP events; prints
TFLOP/s per shape (algorithmic FLOPs) — the tuning harness behind DESIGN.md §4.1.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import vatl_hip as vh  # noqa: E402

SHAPES = {  # name: (H, W, Cin, Cout, k, stride, residual)
    "stem": (256, 192, 3, 64, 7, 2, False),
    "l1.c1s": (64, 48, 64, 64, 1, 1, False), "l2.c1w": (64, 48, 256, 128, 1, 1, False),
    "l1.c1": (64, 48, 256, 64, 1, 1, False), "l1.c2": (64, 48, 64, 64, 3, 1, False), "l1.c3": (64, 48, 64, 256, 1, 1, True),
    "l2.c1": (32, 24, 512, 128, 1, 1, False), "l2.c2": (32, 24, 128, 128, 3, 1, False), "l2.c3": (32, 24, 128, 512, 1, 1, True),
    "l3.c1": (16, 12, 1024, 256, 1, 1, False), "l3.c2": (16, 12, 256, 256, 3, 1, False), "l3.c3": (16, 12, 256, 1024, 1, 1, True),
    "l4.c1": (8, 6, 2048, 512, 1, 1, False), "l4.c2": (8, 6, 512, 512, 3, 1, False), "l4.c3": (8, 6, 512, 2048, 1, 1, True),
    "l3.c2s2": (32, 24, 256, 256, 3, 2, False),
    "l2.c3nr": (32, 24, 128, 512, 1, 1, False), "l3.c3nr": (16, 12, 256, 1024, 1, 1, False), "l1.c3nr": (64, 48, 64, 256, 1, 1, False),
    "deconv1": (8, 6, 2048, 256, 0, 0, False), "deconv3": (32, 24, 256, 256, 0, 0, False),
    "head": (64, 48, 256, 17, 1, 1, False),
    # FastPose-R152 at 384x288 (BASELINE.json configs[4]; use --batch 32): stage-3 / stage-4 bottleneck convs
    "r152.l3.c1": (24, 18, 1024, 256, 1, 1, False), "r152.l3.c2": (24, 18, 256, 256, 3, 1, False), "r152.l3.c3": (24, 18, 256, 1024, 1, 1, True),
    "r152.l2.c2": (48, 36, 128, 128, 3, 1, False), "r152.l4.c2": (12, 9, 512, 512, 3, 1, False), "r152.l4.c1": (12, 9, 2048, 512, 1, 1, False),
    # HRNet-W32 branches (basic blocks: 3x3, residual on the second conv), the 1x1 up paths and the 32->17 head
    "hr.b32": (64, 48, 32, 32, 3, 1, True), "hr.b64": (32, 24, 64, 64, 3, 1, True), "hr.b128": (16, 12, 128, 128, 3, 1, True),
    "hr.b256": (8, 6, 256, 256, 3, 1, True), "hr.up64_32": (32, 24, 64, 32, 1, 1, False), "hr.down32_64": (64, 48, 32, 64, 3, 2, False),
    "hr.head": (64, 48, 32, 17, 1, 1, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", default="")
    ap.add_argument("--vars", default="", help="comma list of k-loop schedule variants to A/B (vatl_tune_set knob 0)")
    ap.add_argument("--ablate", default="", help="comma list of conv ablation bit sets (vatl_tune_set knob 6; needs VATL_ALLOW_ABLATION=1)")
    ap.add_argument("--bm", default="", help="comma list of block-tile row counts to A/B (vatl_tune_set knob 5: 0 auto, 64, 128)")
    ap.add_argument("--splitk", type=int, default=0, help="register an N MB split-K workspace (vatl_set_splitk_workspace) before timing")
    ap.add_argument("--streamk", default="", help="comma list of 0/1: run the layers without / with the stream-K route (vatl_hip.streamk_scope)")
    ap.add_argument("--pdist", default="", help="comma list of operand look-ahead distances of the persistent 1x1 kernel to A/B (vatl_tune_set knob 10: 1, 2)")
    ap.add_argument("--persist", default="", help="comma list of persistent-1x1 settings to A/B (vatl_tune_set knob 7: 0 off, 1 = K <= 256 [default])")
    a = ap.parse_args()
    if a.splitk:
        vh.enable_splitk(a.splitk)
    if os.environ.get("VATL_HALO") == "0":
        vh.tune_set(8, 0)                                  # 32-channel 3x3 layers on the generic kernel (A/B against csrc/conv3x3_halo.hip)
    # clock / cache warm-up: the first configuration measured in a fresh process otherwise reads 4-10 % slow, which biases every A/B
    warm_a = torch.randn((4096, 4096), device="cuda:0")
    t_end = torch.cuda.Event(enable_timing=True); t_beg = torch.cuda.Event(enable_timing=True)
    t_beg.record()
    for _ in range(200):
        warm_a @ warm_a
    t_end.record(); torch.cuda.synchronize()
    del warm_a
    if a.streamk:
        for v in a.streamk.split(","):
            print(f"--- stream-K route {'on' if int(v) else 'off'}")
            if int(v):
                with vh.streamk_scope(torch.device("cuda:0"), force=True):
                    run(a)
            else:
                run(a)
        return
    if a.pdist:
        for v in a.pdist.split(","):
            print(f"--- persistent 1x1 kernel look-ahead {v} k-tile(s)")
            vh.tune_set(10, int(v))
            run(a)
        vh.tune_set(10, 2)
        return
    if a.persist:
        for v in a.persist.split(","):
            print(f"--- persistent 1x1 kernel for K <= 256 * {v}")
            vh.tune_set(7, int(v))
            run(a)
        vh.tune_set(7, 1)
        return
    if a.ablate:
        for v in a.ablate.split(","):
            print(f"--- ablation bits {v} (1 = no epilogue, 2 = one k-tile)")
            vh.tune_set(6, int(v))
            run(a)
        vh.tune_set(6, 0)
        return
    if a.bm:
        for v in a.bm.split(","):
            print(f"--- block tile rows {v}")
            vh.tune_set(5, int(v))
            run(a)
        return
    if a.vars:
        for v in a.vars.split(","):
            parts = (v.split(":") + ["0", "0"])[:3]
            var, order, stag = parts
            print(f"--- schedule variant {var} tile-order {order} stagger {stag}%")
            vh.tune_set(0, int(var))
            vh.tune_set(1, int(order))
            vh.tune_set(2, int(stag))
            run(a)
        return
    run(a)


def run(a):
    dev = torch.device("cuda:0")
    names = a.layers.split(",") if a.layers else [n for n in SHAPES if not n.startswith("hr.") and not n.startswith("r152.") and not n.endswith("nr")]
    if a.layers == "hrnet":
        names = [n for n in SHAPES if n.startswith("hr.")]
    tot_f = tot_t = 0.0
    for name in names:
        H, W, cin, cout, k, stride, res = SHAPES[name]
        B = a.batch
        if k == 0:                                   # deconv 4x4/2
            x = torch.randn((B, H, W, cin), device=dev)
            w = vh.pack_deconv_weight(torch.randn((cin, cout, 4, 4), device=dev) * 0.01)
            sc = torch.ones(cout, device=dev); bi = torch.zeros(cout, device=dev)
            fn = lambda: vh.deconv4x4s2_fwd(x, w, sc, bi, cout, True)
            flops = 2.0 * B * H * W * 16 * cin * cout
        else:
            cpad = 4 if cin == 3 else cin
            x = torch.randn((B, H, W, cpad), device=dev)
            w = vh.pack_conv_weight(torch.randn((cout, cin, k, k), device=dev) * 0.01)
            sc = torch.ones(cout, device=dev); bi = torch.zeros(cout, device=dev)
            Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
            r = torch.randn((B, Ho, Wo, cout), device=dev) if res else None
            nchw = name.endswith("head")
            fn = lambda: vh.conv2d_fwd(x, w, sc, bi, cout, k, k, stride, k // 2, True, residual=r, out_nchw=nchw)
            flops = 2.0 * B * Ho * Wo * cin * cout * k * k
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0.record()
        for _ in range(2):
            fn()
        w1.record(); torch.cuda.synchronize()
        for _ in range(int(min(200, 60.0 / max(w0.elapsed_time(w1) / 2, 1e-3)))):   # >= ~60 ms of the same launch: clocks ramp from the idle state
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        tot_f += flops; tot_t += ms
        print(f"{name:10s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TF/s", flush=True)
    print(f"sum: {tot_t:.2f} ms, {tot_f / tot_t / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
