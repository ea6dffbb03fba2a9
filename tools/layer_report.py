#!/usr/bin/env python3
"""Per-layer TFLOP/s of the SimplePose-R50 conv launches from a rocprofv3 kernel trace.

    python tools/layer_report.py gpurun_out/prof/r1_kernel_trace.csv [batch]

Takes the conv_igemm dispatches of the LAST forward (57, or 53 when conv3 + projection are fused) and pairs them, in launch
order, with the layer list of the network.
"""
import csv
import sys


def layers(B, fused=False, chained=False):
    """fused: the first block of every stage runs conv3 + projection as one dual-source GEMM (K = C1 + C2).
    chained: in stage 1 the second block's conv3 (+ skip) and the third block's conv1 are one launch (bottleneck_chain_kernel), the third block's conv3 its first-GEMM-only form."""
    L = []

    def conv(name, H, W, cin, cout, k, s, res=False):
        Ho, Wo = H // s, W // s
        # compulsory HBM bytes: input pixels that are read (a strided 1x1 reads every other pixel), output, residual, weights
        rd_in = B * (Ho * Wo if k == 1 else H * W) * max(cin, 4) * 4
        byts = rd_in + B * Ho * Wo * cout * 4 * (2 if res else 1) + cin * cout * k * k * 4
        L.append((name, 2 * B * Ho * Wo * cin * cout * k * k, B * Ho * Wo, cout, cin * k * k, byts))
    conv("stem", 256, 192, 3, 64, 7, 2)
    H, W, cin = 64, 48, 64
    for si, (n, wd) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
        for b in range(n):
            s = 2 if (b == 0 and si > 0) else 1
            tag = f"l{si + 1}.{'0' if b == 0 else 'n'}"
            link = chained and si == 0 and b == 1                      # this block's conv3 also computes the next block's conv1
            if not (chained and si == 0 and b == 2):
                conv(tag + ".c1", H, W, cin, wd, 1, 1)
            conv(tag + ".c2", H, W, wd, wd, 3, s)
            if b == 0 and not fused:
                conv(tag + ".proj", H, W, cin, 4 * wd, 1, s)
            Hi, Wi = H, W
            H, W = H // s, W // s
            if b == 0 and fused:
                M = B * H * W
                L.append((tag + ".c3+p", 2 * M * (wd + cin) * 4 * wd, M, 4 * wd, wd + cin, (M * (wd + cin) + M * 4 * wd + (wd + cin) * 4 * wd) * 4))
            elif link:
                M = B * H * W
                L.append((tag + ".c3>c1", 2 * M * wd * 4 * wd * 2, M, 4 * wd, wd, (M * (wd + 4 * wd * 2 + wd) + 2 * wd * 4 * wd) * 4))
            else:
                conv(tag + ".c3", H, W, wd, 4 * wd, 1, 1, res=True)
            cin = 4 * wd
    for i, (ci, co) in enumerate(((2048, 256), (256, 256), (256, 256))):
        L.append((f"deconv{i + 1}", 2 * B * H * W * 16 * ci * co, B * H * W, co, 4 * ci, B * H * W * (ci + 4 * co) * 4 + 16 * ci * co * 4))
        H, W = 2 * H, 2 * W
    conv("head", H, W, 256, 17, 1, 1)
    return L


def main():
    path = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rows = [r for r in csv.DictReader(open(path)) if "conv_igemm" in r["Kernel_Name"] or "gemm1x1_persistent" in r["Kernel_Name"] or "winograd_kernel" in r["Kernel_Name"] or "winograd_persist_kernel" in r["Kernel_Name"] or "winograd_f4_kernel" in r["Kernel_Name"] or "stem_pool_kernel" in r["Kernel_Name"] or "bottleneck_chain_kernel" in r["Kernel_Name"] or "conv1x1_rows_kernel" in r["Kernel_Name"] or "conv1x1_rows256_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    import re
    is_stem = lambda name: bool(re.search(r"conv_igemm_kernel<\d+, \d+, \d+, \d+, true", name)) or "stem_pool_kernel" in name   # (the fused stem launch includes bn1 + relu + maxpool)
    per_step = sum(1 for r in rows if is_stem(r["Kernel_Name"]))                                     # one stem launch per forward
    chained = any("bottleneck_chain_kernel" in r["Kernel_Name"] for r in rows)
    fused = per_step > 0 and len(rows) // per_step == (52 if chained else 53)
    L = layers(B, fused, chained)
    last = rows[-len(L):]
    agg = {}
    for (name, fl, M, N, K, by), r in zip(L, last):
        du = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault((name, M, N, K), [0, 0, 0, 0, False])
        a[0] += fl; a[1] += du; a[2] += 1; a[3] += by
        a[4] = 4 if "winograd_f4" in r["Kernel_Name"] else (2 if "winograd" in r["Kernel_Name"] else 0)
    tot = sum(a[1] for a in agg.values())
    totf = sum(a[0] for a in agg.values())
    print(f"{'layer':12s} {'M':>8s} {'N':>5s} {'K':>5s}  cnt   us/launch   TF/s   share%  mfma_ms   hbm_ms  (ideal at 157.3 TFLOP/s / compulsory bytes at 8 TB/s)")
    bound = 0.0
    for (name, M, N, K), (fl, du, c, by, wino) in agg.items():
        # Winograd launches (W: F(2x2,3x3) / F(3x3,2x2); W4: F(4x4,3x3)): TF/s counts the direct-sum FLOPs; the MFMA floor counts the 16 / 36 (36 / 144) multiplies they execute
        t_m, t_h = fl / 157.3e12 * 1e3 * (0.25 if wino == 4 else (4 / 9 if wino else 1)), by / 8e12 * 1e3
        bound += max(t_m, t_h)
        print(f"{name:12s} {M:8d} {N:5d} {K:5d}  x{c:<2d} {du / c / 1e3:10.1f} {fl / du / 1e3:7.1f} {100 * du / tot:7.2f} {t_m:8.3f} {t_h:8.3f}{'  W4' if wino == 4 else ('  W' if wino else '')}{'  <- HBM-bound' if t_h > t_m else ''}")
    print(f"total {tot / 1e6:.2f} ms for {B} frames -> {totf / tot / 1e3:.1f} TF/s, {B / (tot / 1e9):.0f} frames/s (conv only)")
    print(f"layer-by-layer roofline (each launch at max(MFMA, HBM) time): {bound:.2f} ms -> {B / bound * 1e3:.0f} frames/s; "
          f"measured / that bound = {bound / (tot / 1e6):.3f}")


if __name__ == "__main__":
    main()
