#!/usr/bin/env python3
"""Small-batch latency of the SimplePose-R50 forward (BASELINE.json configs[0] shape, B = 4): eager launches
through ctypes vs one captured HIP graph replay.   python tools/latency_bench.py [--batch 4]"""
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
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--splitk", type=int, default=0, help="megabytes of split-K workspace (vatl_hip.enable_splitk); 0 = off")
    ap.add_argument("--no-guard", action="store_true", help="hip_engine.PARAM_GUARD = False (no parameter-checksum launch in front of the plan calls)")
    a = ap.parse_args()
    if a.no_guard:
        from alphapose.models import hip_engine
        hip_engine.PARAM_GUARD = False
    if a.splitk:
        import vatl_hip as vh
        vh.enable_splitk(a.splitk)
    from bench import build_model
    from active_learning.scoring import score_batch
    dev = torch.device("cuda:0")
    m = build_model(dev)
    x = torch.rand((a.batch, 3, 256, 192), device=dev) - 0.45
    bb = torch.tensor([[100.0, 50.0, 200.0, 183.3]] * a.batch, device=dev)

    def fwd():
        with torch.no_grad():
            hm = m(x)
            return score_batch(hm, bb).keypoints
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fwd()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / a.iters
    res = {"batch": a.batch, "splitk_MB": a.splitk, "param_guard": not a.no_guard, "eager_ms": round(eager * 1e3, 3)}
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fwd()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = fwd()
        g.replay(); torch.cuda.synchronize()
        ref = fwd()
        g.replay(); torch.cuda.synchronize()
        res["graph_equal"] = bool(torch.equal(out, ref))
        t0 = time.perf_counter()
        for _ in range(a.iters):
            g.replay()
        torch.cuda.synchronize()
        res["graph_ms"] = round((time.perf_counter() - t0) / a.iters * 1e3, 3)
    except Exception as e:  # noqa: BLE001
        res["graph_error"] = repr(e)[:300]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
