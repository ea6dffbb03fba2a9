#!/usr/bin/env python3
"""Does running the two halves of a 1024-frame forward on two HIP streams fill the launch tails / overlap HBM-bound with MFMA-bound layers?
One stream x 1024 frames vs S streams x (1024 / S) frames, same kernels (a crop's bits do not depend on its batch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from alphapose.models import hip_engine
    dev = torch.device("cuda:0")
    which = sys.argv[1] if len(sys.argv) > 1 else "r50"
    m = bench.build_model(dev) if which == "r50" else bench.build_net(bench.HRNET_W32, (256, 192), dev).eval()
    n = 1024
    x = torch.rand((n, 3, 256, 192), device=dev) - 0.45
    hm = torch.empty((n, 17, 64, 48), device=dev)
    ref = torch.empty_like(hm)
    with torch.no_grad():
        hip_engine.forward_into(m, x, ref)
    torch.cuda.synchronize()

    def run(s):
        if s == 1:
            hip_engine.forward_into(m, x, hm)
            return
        main_s = torch.cuda.current_stream()
        step = n // s
        for k, st in enumerate(streams[:s]):
            st.wait_stream(main_s)
            with torch.cuda.stream(st):
                hip_engine.forward_into(m, x[k * step:(k + 1) * step], hm[k * step:(k + 1) * step])
        for st in streams[:s]:
            main_s.wait_stream(st)

    streams = [torch.cuda.Stream() for _ in range(4)]
    with torch.no_grad():
        for s in (1, 2, 4, 1, 2, 4):
            run(s); torch.cuda.synchronize()
            same = bool(torch.equal(hm, ref))
            t0 = time.perf_counter()
            for _ in range(5):
                run(s)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            print(f"{which}: {s} stream(s): {dt * 1e3:7.2f} ms per 1024 frames = {n / dt:8.0f} frames/s   bit-identical: {same}", flush=True)


if __name__ == "__main__":
    main()
