#!/usr/bin/env python3
"""Fine-tune step benchmark (BASELINE.json configs[2]): SimpleBaseline-R50,
B = 120 crops, forward in train mode + masked MSE + backward + AdamW (3 param groups).

    python tools/train_bench.py [--batch 120] [--steps 5] [--warmup 4]
Prints one JSON line: steps/s, crops/s and the conv-path TFLOP/s (3 x 10.853 GFLOP per crop).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=120)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--wgrad-blocks", type=int, default=0, help="vatl_tune_set(3, v): target block count of the wgrad launches")
    ap.add_argument("--single-stream", action="store_true", help="weight gradients on the main stream (profiling: per-kernel durations add up to the step)")
    ap.add_argument("--model", default="simplepose", choices=["simplepose", "fastpose", "hrnet"], help="backbone (autograd path for fastpose / hrnet)")
    a = ap.parse_args()
    import vatl_hip as vh
    if a.wgrad_blocks:
        vh.tune_set(3, a.wgrad_blocks)
    from active_learning.optim import AdamW
    from alphapose.models import builder
    from alphapose.utils.config import edict
    dev = torch.device("cuda:0")
    cfg = edict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
    if a.model == "fastpose":
        cfg = edict({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50})
    if a.model == "hrnet":
        cfg = edict({"TYPE": "PoseHighResolutionNet", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50, "FINAL_CONV_KERNEL": 1, "PRETRAINED_LAYERS": ["*"],
                     "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "NUM_BLOCKS": [4, 4], "NUM_CHANNELS": [32, 64], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
                     "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "NUM_BLOCKS": [4, 4, 4], "NUM_CHANNELS": [32, 64, 128], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
                     "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "NUM_BLOCKS": [4, 4, 4, 4], "NUM_CHANNELS": [32, 64, 128, 256], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"}})
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    torch.manual_seed(166)
    m = builder.build_sppe(cfg, preset_cfg=preset).to(dev).train()
    lr = 2.5e-4
    if a.model == "simplepose":
        opt = AdamW(params=[{"params": m.final_layer.parameters(), "lr": lr * 10}, {"params": m.preact.parameters(), "lr": lr},
                            {"params": m.deconv_layers.parameters(), "lr": lr * 5}], weight_decay=0.7)
    else:
        opt = AdamW(params=[{"params": m.parameters(), "lr": lr}], weight_decay=0.7)
    g = torch.Generator(device=dev); g.manual_seed(166)
    x = torch.rand((a.batch, 3, 256, 192), device=dev, generator=g) - 0.45
    labels = torch.rand((a.batch, 17, 64, 48), device=dev, generator=g) * 0.1
    masks = (torch.rand((a.batch, 17, 1, 1), device=dev, generator=g) > 0.2).float()
    from alphapose.models import hip_train
    if a.single_stream:
        hip_train._side.enabled = False

    def step_autograd():                                           # the module's own autograd path (FastPose / HRNet trainers behind it)
        opt.zero_grad(set_to_none=True)
        out = m(x)
        loss, dout = vh.masked_mse_fwd_bwd(out.detach(), labels, masks)
        out.backward(dout)
        opt.step()
        return loss

    def step():
        if a.model != "simplepose":
            return step_autograd()
        tr = hip_train.trainer_for(m)
        with torch.no_grad():
            out = tr.forward(x)
            loss, dout = vh.masked_mse_fwd_bwd(out, labels, masks)       # fused loss + gradient
            grads = tr.backward(dout)
            for p_, g_ in grads.items():
                p_.grad = g_
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": f"fine-tune steps/s (fwd+bwd+AdamW), {a.model} 256x192", "batch": a.batch, "ms_per_step": round(dt * 1e3, 2),
                      "crops_per_s": round(a.batch / dt, 1), "conv_tflops": round(3 * {"simplepose": 10.853e9, "fastpose": 11.774e9, "hrnet": 15.29e9}[a.model] * a.batch / dt / 1e12, 2),
                      "loss": float(loss), "wgrad_blocks": a.wgrad_blocks}))


if __name__ == "__main__":
    main()
