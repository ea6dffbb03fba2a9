#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json / SURVEY.md §8(d) on ONE MI355X (the headline metric is bench.py):

  cfg3  SimpleBaseline-R50: 10 fine-tune steps (fwd + masked MSE + bwd + AdamW x{10,1,5}, wd 0.7) at B = 120, then
        inference + THC-L1 over a 1024-frame video, both "reference-faithful" (prev / current / next crops forwarded
        for every item, ActiveLearning.py:277,294,296) and de-duplicated (one forward per frame)
  cfg4  HRNet-W32 inference + THC-L1 + WPU (42-d auto-encoder, z = 4) on one rank's 1024-frame shard
  cfg5  FastPose-R152 384x288 fine-tune step at B = 32 (per-GPU share of the DP batch; the 301 MB gradient
        all-reduce is not part of a 1-GPU run)

    python tools/config_bench.py [--only cfg3,cfg4,cfg5]
One JSON line per measurement.
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

HRNET = {"TYPE": "PoseHighResolutionNet", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50, "FINAL_CONV_KERNEL": 1, "PRETRAINED_LAYERS": ["*"],
         "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "NUM_BLOCKS": [4, 4], "NUM_CHANNELS": [32, 64], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
         "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "NUM_BLOCKS": [4, 4, 4], "NUM_CHANNELS": [32, 64, 128], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
         "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "NUM_BLOCKS": [4, 4, 4, 4], "NUM_CHANNELS": [32, 64, 128, 256], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"}}


def build(cfg, hw, dev):
    from alphapose.models import builder
    from alphapose.utils.config import edict
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": list(hw), "HEATMAP_SIZE": [hw[0] // 4, hw[1] // 4]})
    torch.manual_seed(166)
    m = builder.build_sppe(edict(cfg), preset_cfg=preset)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
    return m.to(dev)


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def train_step_fn(m, opt, x, labels, masks):
    import vatl_hip as vh
    from alphapose.models import hip_train

    def step():
        tr = hip_train.trainer_for(m)
        with torch.no_grad():
            out = tr.forward(x)
            loss, dout = vh.masked_mse_fwd_bwd(out, labels, masks)
            for p_, g_ in tr.backward(dout).items():
                p_.grad = g_
        opt.step()
        return loss
    return step


def video(dev, n, hw, tracks=4):
    g = torch.Generator(device=dev); g.manual_seed(166)
    x = torch.rand((n, 3, hw[0], hw[1]), device=dev, generator=g) - torch.tensor([0.406, 0.457, 0.480], device=dev).view(1, 3, 1, 1)
    w = 60 + 180 * torch.rand(n, device=dev, generator=g)
    bbox = torch.stack([torch.full_like(w, 100.0), torch.full_like(w, 50.0), 100 + w, 50 + w * 4 / 3], 1).contiguous()
    pos = torch.arange(n, device=dev) % (n // tracks)
    return x, bbox, (pos != 0).to(torch.uint8), (pos != n // tracks - 1).to(torch.uint8)


def cfg3(dev):
    import vatl_hip as vh
    from active_learning.optim import AdamW
    from active_learning.scoring import score_batch
    from alphapose.models import hip_engine
    m = build({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50}, (256, 192), dev).train()
    lr = 2.5e-4
    opt = AdamW(params=[{"params": m.final_layer.parameters(), "lr": lr * 10}, {"params": m.preact.parameters(), "lr": lr},
                        {"params": m.deconv_layers.parameters(), "lr": lr * 5}], weight_decay=0.7)
    g = torch.Generator(device=dev); g.manual_seed(166)
    b = 120
    x = torch.rand((b, 3, 256, 192), device=dev, generator=g) - 0.45
    labels = torch.rand((b, 17, 64, 48), device=dev, generator=g) * 0.1
    masks = (torch.rand((b, 17, 1, 1), device=dev, generator=g) > 0.2).float()
    dt = timed(train_step_fn(m, opt, x, labels, masks), 10, 4)     # four warm-up steps: see bench.py extra_finetune
    print(json.dumps({"config": "cfg3 fine-tune step, SimpleBaseline-R50 B=120", "ms_per_step": round(dt * 1e3, 2), "crops_per_s": round(b / dt, 1),
                      "conv_tflops": round(3 * 10.853e9 * b / dt / 1e12, 2)}), flush=True)
    m.eval()
    n = 1024
    vx, bbox, ip, inx = video(dev, n, (256, 192))
    hm = torch.empty((n, 17, 64, 48), device=dev)

    def dedup():
        with torch.no_grad():
            hip_engine.forward_into(m, vx, hm)
            return score_batch(hm, bbox, ip, inx, thc_norm="L1")

    prev_x = torch.cat([torch.zeros_like(vx[:1]), vx[:-1]]) * ip.view(-1, 1, 1, 1)
    next_x = torch.cat([vx[1:], torch.zeros_like(vx[:1])]) * inx.view(-1, 1, 1, 1)
    hp, hn = torch.empty_like(hm), torch.empty_like(hm)

    def faithful():                                   # three forwards per item, THC from explicit neighbour heat-maps
        with torch.no_grad():
            hip_engine.forward_into(m, vx, hm)
            hip_engine.forward_into(m, prev_x, hp)
            hip_engine.forward_into(m, next_x, hn)
            s = score_batch(hm, bbox, ip, inx, thc_norm=None)
            tp, tn = vh.thc_pairs(hm, hp, "L1"), vh.thc_pairs(hm, hn, "L1")
            one = (ip ^ inx).float()
            s.thc = (tp * ip + tn * inx) * (1 + one)
            return s

    a, bb = dedup(), faithful()
    torch.cuda.synchronize()
    same = bool(torch.equal(a.thc, bb.thc))
    d1, d3 = timed(dedup, 3, 1), timed(faithful, 3, 1)
    print(json.dumps({"config": "cfg3 inference + THC-L1, 1024-frame video", "dedup_frames_per_s": round(n / d1, 1), "faithful_3fwd_frames_per_s": round(n / d3, 1),
                      "thc_bit_identical": same}), flush=True)


def cfg4(dev):
    from active_learning.scoring import score_batch
    from active_learning.Whole_body_AE.AutoEncoder import WholeBodyAE
    from alphapose.models import hip_engine
    m = build(HRNET, (256, 192), dev).eval()
    ae = WholeBodyAE(z_dim=4, kp_direct=False, input_dim=42).to(dev)
    n = 1024
    vx, bbox, ip, inx = video(dev, n, (256, 192))
    hm = torch.empty((n, 17, 64, 48), device=dev)
    flat = ae.packed()

    def run():
        with torch.no_grad():
            hip_engine.forward_into(m, vx, hm)
            return score_batch(hm, bbox, ip, inx, thc_norm="L1", ae_flat=flat, ae_dims=(42, 4))

    dt = timed(run, 3, 1)
    print(json.dumps({"config": "cfg4 HRNet-W32 inference + THC-L1 + WPU, 1024-frame shard (1 rank)", "frames_per_s": round(n / dt, 1),
                      "conv_tflops": round(15.29e9 * n / dt / 1e12, 2)}), flush=True)


def cfg5(dev):
    from active_learning.optim import AdamW
    m = build({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 152}, (384, 288), dev).train()
    lr = 2.5e-4
    opt = AdamW(params=[{"params": m.conv_out.parameters(), "lr": lr * 10}, {"params": m.preact.parameters(), "lr": lr},
                        {"params": m.duc1.parameters(), "lr": lr * 5}, {"params": m.duc2.parameters(), "lr": lr * 5}], weight_decay=0.7)
    g = torch.Generator(device=dev); g.manual_seed(166)
    b = 32
    x = torch.rand((b, 3, 384, 288), device=dev, generator=g) - 0.45
    labels = torch.rand((b, 17, 96, 72), device=dev, generator=g) * 0.1
    masks = (torch.rand((b, 17, 1, 1), device=dev, generator=g) > 0.2).float()
    dt = timed(train_step_fn(m, opt, x, labels, masks), 5, 4)
    print(json.dumps({"config": "cfg5 fine-tune step, FastPose-R152 384x288 B=32 (one rank's share)", "ms_per_step": round(dt * 1e3, 2),
                      "crops_per_s": round(b / dt, 1), "conv_tflops": round(3 * 59.192e9 * b / dt / 1e12, 2),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)


def fastpose_infer(dev):
    """FastPose inference (not one of BASELINE.json's configs; here to keep an eye on the SE / DUC path)."""
    from active_learning.scoring import score_batch
    from alphapose.models import hip_engine
    for layers, hw, n, gflop in ((50, (256, 192), 1024, 11.774), (152, (384, 288), 256, 59.192)):
        m = build({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": layers}, hw, dev).eval()
        vx, bbox, ip, inx = video(dev, n, hw)
        hm = torch.empty((n, 17, hw[0] // 4, hw[1] // 4), device=dev)

        def run():
            with torch.no_grad():
                hip_engine.forward_into(m, vx, hm)
                return score_batch(hm, bbox, ip, inx, thc_norm="L1")
        dt = timed(run, 3, 1)
        print(json.dumps({"config": f"FastPose-R{layers} {hw[0]}x{hw[1]} inference + THC-L1, {n} frames", "frames_per_s": round(n / dt, 1),
                          "conv_tflops": round(gflop * 1e9 * n / dt / 1e12, 2)}), flush=True)
        del m, vx, hm
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="cfg3,cfg4,cfg5")
    ap.add_argument("--single-stream", action="store_true", help="weight gradients on the main stream (profiling: per-kernel durations add up to the step)")
    a = ap.parse_args()
    if a.single_stream:
        from alphapose.models import hip_train
        hip_train._side.enabled = False
    dev = torch.device("cuda:0")
    for name in a.only.split(","):
        {"cfg3": cfg3, "cfg4": cfg4, "cfg5": cfg5, "fastpose": fastpose_infer}[name](dev)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
