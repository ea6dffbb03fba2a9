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
    cases = [
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
