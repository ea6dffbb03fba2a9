#!/usr/bin/env python3
"""HBM roofline of the heat-map scorers (decode, THC, local-peak, WPU, TPC, MPE/Margin, Entropy, masked MSE, AdamW).

    python tools/scorer_bench.py [--items 4096] [--iters 10]
Heat-maps of 4096 items are 856 MB: larger than the 256 MB Infinity Cache, so repeated launches stream from HBM.
Prints one JSON line per kernel: us per launch, algorithmic bytes, GB/s, fraction of the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="", help="substring of the kernel name to run alone")
    a = ap.parse_args()
    import vatl_hip as vh
    from active_learning.Whole_body_AE.AutoEncoder import WholeBodyAE
    dev = torch.device("cuda:0")
    n, J, H, W = a.items, 17, 64, 48
    g = torch.Generator(device=dev); g.manual_seed(1)
    hm = torch.rand((n, J, H, W), device=dev, generator=g)
    hm2 = torch.rand((n, J, H, W), device=dev, generator=g)
    w = 60 + 180 * torch.rand(n, device=dev, generator=g)
    bbox = torch.stack([torch.full_like(w, 100.0), torch.full_like(w, 50.0), 100 + w, 50 + w * 4 / 3], 1).contiguous()
    ip = torch.ones(n, dtype=torch.uint8, device=dev); ip[0] = 0
    inx = torch.ones(n, dtype=torch.uint8, device=dev); inx[-1] = 0
    plane = J * H * W * 4
    coords, maxv, _ = vh.decode(hm, bbox)
    kp = torch.cat([coords, maxv.unsqueeze(-1)], 2).contiguous()
    ae = WholeBodyAE(z_dim=4, kp_direct=False, input_dim=42).to(dev).packed()
    mask = (torch.rand((n, J, 1, 1), device=dev, generator=g) > 0.2).float()
    npar = 34_000_000
    p_, g_, m_, v_ = (torch.rand(npar, device=dev) for _ in range(4))
    # crop producer (§8f rank 2): 32 synthetic 1080p frames (199 MB), n person boxes 60-240 px wide -> (n,3,256,192) fp32
    import numpy as np
    from alphapose.utils.bbox import box_to_center_scale_batch
    from alphapose.utils.transforms import get_affine_transform_batch, invert_affine_batch
    arena = torch.randint(0, 256, (32 * 1080 * 1920 * 3,), device=dev, dtype=torch.uint8, generator=g)
    rs = np.random.RandomState(2)
    bw = rs.uniform(60, 240, n); bx, by = rs.uniform(0, 1600, n), rs.uniform(0, 700, n)
    cc, ss = box_to_center_scale_batch(np.stack([bx, by, bx + bw, by + bw * 4 / 3], 1), 0.75)
    minv = torch.from_numpy(invert_affine_batch(get_affine_transform_batch(cc, ss, 0.0, [192, 256]))).to(dev)
    fidx = rs.randint(0, 32, n)
    soff = torch.from_numpy(fidx.astype(np.int64) * (1080 * 1920 * 3)).to(dev)
    hwf = torch.tensor([[1080, 1920, 0]], dtype=torch.int32, device=dev).repeat(n, 1).contiguous()
    crop_out = torch.empty((n, 3, 256, 192), device=dev)
    crop_bytes = int(n * 3 * 256 * 192 * 4 + (ss[:, 0].astype(np.float64) * ss[:, 1]).sum() * 3)        # output + the source window once
    cases = [
        ("crop_warp_affine (f2: warpAffine + im_to_torch + mean)", lambda: vh.crop_warp_affine(arena, soff, hwf, minv, (256, 192), out=crop_out), crop_bytes),
        ("decode_argmax_affine (a8)", lambda: vh.decode(hm, bbox), n * (plane + 204)),
        ("thc_stream L1 (a9)", lambda: vh.thc_stream(hm, ip, inx, "L1"), n * 2 * plane),
        ("thc_pairs L1 (a9, explicit neighbour maps)", lambda: vh.thc_pairs(hm, hm2, "L1"), n * 2 * plane),
        ("localpeak_mean (a11)", lambda: vh.localpeak_mean(hm), n * (plane + 4)),
        ("hybrid_ae_wpu (a12)", lambda: vh.hybrid_ae_wpu(kp, bbox, ae, 42, 4), n * 224),
        ("tpc_stream (a10)", lambda: vh.tpc_stream(hm, bbox, coords, ip, inx), n * 2 * plane),
        ("decode_softargmax (a8')", lambda: vh.decode_softargmax(hm, bbox, "softmax"), n * plane),
        ("peaks5 MPE/Margin (a13)", lambda: vh.peaks5(hm, 5), n * plane),
        ("plane_entropy (a13)", lambda: vh.plane_entropy(hm), n * plane),
        ("masked_mse_fwd_bwd (a6)", lambda: vh.masked_mse_fwd_bwd(hm, hm2, mask), n * 3 * plane),
        ("adamw_step 34 M parameters (a7)", lambda: vh.adamw_step(p_, g_, m_, v_, 3, 1e-3, 0.7), npar * 28),
    ]
    for name, fn, byts in cases:
        if a.only and a.only not in name:
            continue
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / a.iters
        print(json.dumps({"kernel": name, "items": n, "us": round(us, 1), "algorithmic_MB": round(byts / 1e6, 1), "GB_per_s": round(byts / us / 1e3, 1),
                          "frac_of_8TBps": round(byts / us / 1e3 / 8000, 3)}), flush=True)


if __name__ == "__main__":
    main()
