#!/usr/bin/env python3
"""Per-call time of every conv / weight-gradient entry point inside ONE fine-tune step of SimpleBaseline-R50 at B = 120 (single stream,
a device synchronize around every call): which layer shapes the step spends its time in.

    python tools/step_calls.py
"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import config_bench as cb
import vatl_hip as vh
from alphapose.models import hip_train
dev = torch.device("cuda:0")
cfg = {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50}
m = cb.build(cfg, (256, 192), dev).train()
from active_learning import optim as O
B = 120
x = torch.randn((B, 3, 256, 192), device=dev)
labels = torch.rand((B, 17, 64, 48), device=dev); masks = torch.ones((B, 17, 1, 1), device=dev)
opt = O.AdamW([{"params": list(m.parameters()), "lr": 1e-4}], weight_decay=0.7)
os.environ["VATL_WGRAD_STREAM"] = "0"
step = cb.train_step_fn(m, opt, x, labels, masks)
for _ in range(3): step()
torch.cuda.synchronize()
rows = []
def wrap(name, fmt):
    orig = getattr(vh, name)
    def inner(*a, **k):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **k); e1.record(); torch.cuda.synchronize()
        rows.append((e0.elapsed_time(e1) * 1e3, name, fmt(a, k)))
        return r
    setattr(vh, name, inner)
sh = lambda t: tuple(t.shape)
wrap("conv2d_fwd_ex", lambda a, k: (sh(a[0]), "cout", a[2], "k", a[3], "s", a[5], "res" if k.get("residual") is not None else ""))
wrap("conv2d_fwd_ex_bnbwd", lambda a, k: (sh(a[0]), "cout", a[2], "k", a[3], "s", a[5], "res" if k.get("residual") is not None else ""))
wrap("conv2d_fwd_bnstats", lambda a, k: (sh(a[0]), "cout", a[2], "k", a[3], "s", a[5]))
wrap("conv2d_fwd", lambda a, k: (sh(a[0]), "cout", a[4], "k", a[5]))
wrap("conv2d_wgrad", lambda a, k: (sh(a[0]), sh(a[1]), a[4:8]))
step(); torch.cuda.synchronize()
tot = collections.defaultdict(lambda: [0, 0.0])
for t, n, f in rows:
    tot[(n, f)][0] += 1; tot[(n, f)][1] += t
for (n, f), (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:28]:
    xs = f[0]; M = xs[0] * xs[1] * xs[2]
    print(f"{t:8.0f} us x{c}  {n} {f}")
