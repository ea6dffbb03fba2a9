#!/usr/bin/env python3
"""Per-shape table of one HRNet-W32 stream pass (every conv / Winograd / fused launch timed with events, grouped by entry point + shape).
usage: hrnet_layer_report.py [frames, default 1024] [model: hrnet | simplepose | fastpose]"""
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vatl_hip as vh  # noqa: E402

NAMES = ["conv2d_fwd", "conv3x3_winograd_fwd", "conv3x3_winograd_f4_fwd", "conv3x3_winograd_c32_fwd", "conv1x1_dual_fwd", "conv1x1_rows_fwd", "fuse_upsample_add", "bottleneck_chain_fwd", "stem3_fwd", "stem_pool_fwd", "deconv4x4s2_winograd_fwd", "deconv4x4s2_fwd",
         "fuse_up", "maxpool3x3s2_fwd", "nchw_to_nhwc", "pixelshuffle2_fwd", "se_scale_add_relu", "gap_fwd"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    kind = sys.argv[2] if len(sys.argv) > 2 else "hrnet"
    from alphapose.models import builder, hip_engine
    from alphapose.utils.config import edict
    dev = torch.device("cuda:0")
    cfgs = {"hrnet": {"TYPE": "PoseHighResolutionNet", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50, "FINAL_CONV_KERNEL": 1, "PRETRAINED_LAYERS": ["*"],
                      "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "NUM_BLOCKS": [4, 4], "NUM_CHANNELS": [32, 64], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
                      "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "NUM_BLOCKS": [4, 4, 4], "NUM_CHANNELS": [32, 64, 128], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
                      "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "NUM_BLOCKS": [4, 4, 4, 4], "NUM_CHANNELS": [32, 64, 128, 256], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"}},
            "simplepose": {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50},
            "fastpose": {"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50}}
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    torch.manual_seed(0)
    m = builder.build_sppe(edict(cfgs[kind]), preset_cfg=preset).to(dev).eval()
    x = torch.randn((n, 3, 256, 192), device=dev)
    out = torch.empty((n, 17, 64, 48), device=dev)
    with torch.no_grad():
        for _ in range(2):
            hip_engine.forward_into(m, x, out)
    torch.cuda.synchronize()
    events = []
    originals = {k: getattr(vh, k) for k in NAMES if hasattr(vh, k)}

    def wrap(name, fn):
        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            t0 = a[0] if isinstance(a[0], torch.Tensor) else a[0][0]
            ro = r[0] if isinstance(r, tuple) else r
            extra = ""
            if name == "conv2d_fwd":
                extra = f" k{a[5]} s{a[7]}" + (" +res" if k.get("residual") is not None else "")
            elif name in ("conv3x3_winograd_fwd", "conv3x3_winograd_f4_fwd"):
                extra = " +res" if k.get("residual") is not None else ""
            events.append((name + extra, tuple(t0.shape[1:]), tuple(ro.shape[1:]), e0, e1))
            return r
        return inner
    for k, fn in originals.items():
        setattr(vh, k, wrap(k, fn))
    try:
        with torch.no_grad(), vh.flop_meter() as fm:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            hip_engine.forward_into(m, x, out)
            b.record()
        torch.cuda.synchronize()
    finally:
        for k, fn in originals.items():
            setattr(vh, k, fn)
    agg = OrderedDict()
    for name, si, so, e0, e1 in events:
        g = agg.setdefault((name, si, so), [0, 0.0])
        g[0] += 1; g[1] += e0.elapsed_time(e1)
    tot = sum(g[1] for g in agg.values())
    print(f"{kind} pass of {n} frames: {a.elapsed_time(b):.2f} ms wall, {tot:.2f} ms in {len(events)} timed launches, executed {fm.total / 1e12:.2f} TFLOP")
    print(f"{'entry point':34s} {'in (H, W, C)':>18s} {'out':>18s}  cnt   us/launch   total ms  share%   alg TF/s")
    for (name, si, so), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fl = 0.0
        if name.startswith("conv2d_fwd") or name.startswith("conv3x3_winograd"):
            kk = 9 if ("winograd" in name or " k3" in name) else (1 if " k1" in name else 0)
            if kk and len(so) == 3 and len(si) == 3:
                fl = 2.0 * n * so[0] * so[1] * so[2] * si[2] * kk
        print(f"{name:34s} {str(si):>18s} {str(so):>18s}  x{c:<3d} {ms / c * 1e3:10.1f} {ms:10.2f} {100 * ms / tot:7.2f} {(fl / (ms / c * 1e-3) / 1e12) if fl else 0:9.1f}")


if __name__ == "__main__":
    main()
