#!/usr/bin/env python3
"""What would a fusion buy?  Upper bounds measured BEFORE building anything: the fine-tune step of configs 3 / 5 with one piece of work
removed at the Python level (the results are then WRONG — only the step time is read), next to the unmodified step.

    python tools/ablate_train_step.py [cfg3|cfg5] [--steps 8]

  base            the step as shipped (two-stream weight gradients)
  no_mask_y       the BatchNorm-backward epilogues of the data gradients do not read the saved block output for the ReLU mask
                  (= what a 1-bit mask tensor would save: one 4C-wide pass per bottleneck)
  no_sba_noskip   scale_bias_act of the layers WITHOUT a skip input returns z (= BatchNorm affine + ReLU folded into the consumer's operand load)
  no_sba          every scale_bias_act returns z / its input (the whole forward element-wise pass)
  no_apply        bn_bwd_from_stats returns g as dz (= the apply pass folded into the consumers)
  no_finalize     the tiny per-layer finalize launches of both directions skipped (stale scale / bias)
  bn_minc64       BatchNorm-backward epilogue also for the 64-channel layers (hip_train._FUSE_BN_MIN_C = 64)
  wwg_minc64/256  Winograd weight gradients from 64 / 256 channels on (hip_train._WINOGRAD_WGRAD_MIN_C)
  one_stream      weight gradients on the main stream
  streamk         stream-K scheduling of the under-filled conv launches (vatl_hip.STREAMK_IN_TRAINING; correct, bit-identical)
  bm64 / bm128    tile rows of the implicit-GEMM forward / data-gradient launches forced (vatl_tune_set(5, v); correct, bit-identical)
  f4_off / f4_on  F(4x4,3x3) for the 3x3 layers' forward + data gradient off / on (NOT an ablation: both are correct; hip_train._WINOGRAD_F4)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
import vatl_hip as vh  # noqa: E402
from alphapose.models import hip_train  # noqa: E402
from active_learning.optim import AdamW  # noqa: E402

which = next((a for a in sys.argv[1:] if a.startswith("cfg")), "cfg3")
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 8
only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
dev = torch.device("cuda:0")
if which == "cfg5":
    cfg, hw, B, groups = bench.FAST_R152, (384, 288), 32, (("conv_out", 10), ("preact", 1), ("duc1", 5), ("duc2", 5))
else:
    cfg, hw, B, groups = bench.SIMPLE_R50, (256, 192), 120, (("final_layer", 10), ("preact", 1), ("deconv_layers", 5))


def run(name, setup, teardown):
    m = bench.build_net(cfg, hw, dev).train()
    opt = AdamW(params=[{"params": getattr(m, a).parameters(), "lr": 2.5e-4 * f} for a, f in groups], weight_decay=0.7)
    g = torch.Generator(device=dev); g.manual_seed(166)
    x = torch.rand((B, 3, hw[0], hw[1]), device=dev, generator=g) - 0.45
    labels = torch.rand((B, 17, hw[0] // 4, hw[1] // 4), device=dev, generator=g) * 0.1
    masks = (torch.rand((B, 17, 1, 1), device=dev, generator=g) > 0.2).float()
    setup()
    try:
        step, _ = bench.finetune_step_fn(m, opt, x, labels, masks, 1)
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps * 1e3)
    finally:
        teardown()
    print(json.dumps({"config": which, "variant": name, "ms_per_step": [round(t, 2) for t in ts], "best": round(min(ts), 2)}), flush=True)
    del m, opt, step
    torch.cuda.empty_cache()


saved = {}


def patch(obj, attr, new):
    saved[(obj, attr)] = getattr(obj, attr)
    setattr(obj, attr, new)


def restore():
    for (obj, attr), v in saved.items():
        setattr(obj, attr, v)
    saved.clear()
    vh.tune_set(5, 0)


def no_mask_y():
    init = vh.BnBwdSpec.__init__

    def inner(self, z, mean, invstd, mask_y=None, scale=None, bias=None):
        init(self, z, mean, invstd, mask_y=None, scale=scale, bias=bias)
    patch(vh.BnBwdSpec, "__init__", inner)


def no_sba(noskip_only):
    orig = vh.scale_bias_act

    def inner(z, scale, bias, residual=None, relu=True):
        if residual is None or not noskip_only:
            return z
        return orig(z, scale, bias, residual, relu)
    patch(vh, "scale_bias_act", inner)


def no_apply():
    def inner(spec, g, gamma, dgamma=None, dbeta=None):
        c = spec.z.shape[-1]
        dgamma = dgamma if dgamma is not None else torch.zeros(c, device=g.device)
        dbeta = dbeta if dbeta is not None else torch.zeros(c, device=g.device)
        return g, dgamma, dbeta
    patch(vh, "bn_bwd_from_stats", inner)


def no_finalize():
    cache = {}

    def inner(stats, nblk, m, c, bn_args, device):
        if c not in cache:
            cache[c] = [torch.ones(c, device=device) * v for v in (0.0, 1.0, 1.0, 0.0)]
        return cache[c]
    patch(vh, "_bn_finalize", inner)


VARIANTS = [
    ("base", lambda: None),
    ("no_mask_y", no_mask_y),
    ("no_sba_noskip", lambda: no_sba(True)),
    ("no_sba", lambda: no_sba(False)),
    ("no_apply", no_apply),
    ("no_finalize", no_finalize),
    ("bn_minc64", lambda: patch(hip_train, "_FUSE_BN_MIN_C", 64)),
    ("wwg_minc64", lambda: patch(hip_train, "_WINOGRAD_WGRAD_MIN_C", 64)),
    ("wwg_minc256", lambda: patch(hip_train, "_WINOGRAD_WGRAD_MIN_C", 256)),
    ("one_stream", lambda: patch(hip_train._side, "enabled", False)),
    ("streamk", lambda: patch(vh, "STREAMK_IN_TRAINING", True)),                  # stream-K scheduling of the under-filled conv launches (bit-identical; off in the product)
    ("bm64", lambda: vh.tune_set(5, 64)), ("bm128", lambda: vh.tune_set(5, 128)),  # forward / data-gradient tile rows forced (product knob 5; 0 = the automatic choice)
    ("f4_off", lambda: patch(hip_train, "_WINOGRAD_F4", False)),
    ("f4_on", lambda: patch(hip_train, "_WINOGRAD_F4", True)),
    ("base", lambda: None),
]
for name, setup in VARIANTS:
    if only and name not in only:
        continue
    run(name, setup, restore)
