#!/usr/bin/env python3
"""EXPERIMENT: the 1024-frame SimplePose-R50 / HRNet stream pass eager vs replayed from a HIP graph (52 / ~300 launches with ~5 - 10 us between them)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from alphapose.models import builder, hip_engine
    from alphapose.utils.config import edict
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hrnet_layer_report as hl  # noqa: F401  (config dicts)
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    cfgs = {"simplepose": {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50},
            "hrnet": {"TYPE": "PoseHighResolutionNet", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50, "FINAL_CONV_KERNEL": 1, "PRETRAINED_LAYERS": ["*"],
                      "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "NUM_BLOCKS": [4, 4], "NUM_CHANNELS": [32, 64], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
                      "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "NUM_BLOCKS": [4, 4, 4], "NUM_CHANNELS": [32, 64, 128], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
                      "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "NUM_BLOCKS": [4, 4, 4, 4], "NUM_CHANNELS": [32, 64, 128, 256], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"}}}
    for kind in ("simplepose", "hrnet"):
        torch.manual_seed(0)
        m = builder.build_sppe(edict(cfgs[kind]), preset_cfg=preset).to(dev).eval()
        x = torch.randn((n, 3, 256, 192), device=dev)
        out = torch.empty((n, 17, 64, 48), device=dev)
        with torch.no_grad():
            for _ in range(3):
                hip_engine.forward_into(m, x, out)
            torch.cuda.synchronize()
            ref = out.clone()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                hip_engine.forward_into(m, x, out)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                hip_engine.forward_into(m, x, out)
            out.zero_()
            g.replay(); torch.cuda.synchronize()
            same = bool(torch.equal(out, ref))

            def timed(fn, it=10):
                fn(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(it):
                    fn()
                b.record(); torch.cuda.synchronize()
                return a.elapsed_time(b) / it
            for rep in range(3):
                te = timed(lambda: hip_engine.forward_into(m, x, out))
                tg = timed(g.replay)
                print(f"{kind} {n} frames: eager {te:.3f} ms  graph replay {tg:.3f} ms  ({te / tg:.4f}x)  bit-identical {same}", flush=True)
        del g


if __name__ == "__main__":
    main()
