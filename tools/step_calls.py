#!/usr/bin/env python3
"""Per-call time of EVERY vatl_hip entry point inside ONE fine-tune step (single stream, a device synchronize around every call):
which layer shapes — convs, weight gradients, BatchNorm passes — the step spends its time in.

    python tools/step_calls.py [cfg3|cfg5] [--top N]
cfg3 = SimpleBaseline-R50 256x192 B = 120, cfg5 = FastPose-R152 384x288 B = 32.
"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import config_bench as cb
import vatl_hip as vh
from alphapose.models import hip_train
hip_train._side.enabled = False          # one stream: every call is timed on its own
from active_learning import optim as O

which = next((a for a in sys.argv[1:] if a.startswith("cfg")), "cfg3")
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
dev = torch.device("cuda:0")
if which == "cfg5":
    cfg = {"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 152}
    hw, B = (384, 288), 32
else:
    cfg = {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50}
    hw, B = (256, 192), 120
m = cb.build(cfg, hw, dev).train()
x = torch.randn((B, 3, hw[0], hw[1]), device=dev)
labels = torch.rand((B, 17, hw[0] // 4, hw[1] // 4), device=dev); masks = torch.ones((B, 17, 1, 1), device=dev)
opt = O.AdamW([{"params": list(m.parameters()), "lr": 1e-4}], weight_decay=0.7)
step = cb.train_step_fn(m, opt, x, labels, masks)
for _ in range(3): step()
torch.cuda.synchronize()
rows = []
SKIP = {"lib", "upload", "tune_set", "set_pack_plan", "latency_mode", "conv_cout_pad", "enable_splitk"}


def describe(a, k):
    out = []
    for v in list(a) + [f"{n}=" for n, v in k.items() if v is not None and not torch.is_tensor(v)] + [v for v in k.values()]:
        if torch.is_tensor(v):
            out.append("x".join(str(d) for d in v.shape))
        elif isinstance(v, (int, bool, str)):
            out.append(str(v))
        elif isinstance(v, vh.BnBwdSpec):
            out.append("spec")
        elif v is None:
            out.append("-")
    return " ".join(out)[:110]


def wrap(name):
    orig = getattr(vh, name)
    def inner(*a, **k):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **k); e1.record(); torch.cuda.synchronize()
        rows.append((e0.elapsed_time(e1) * 1e3, name, describe(a, k)))
        return r
    setattr(vh, name, inner)


for name in dir(vh):
    f = getattr(vh, name)
    if name.startswith("_") or name in SKIP or not callable(f) or isinstance(f, type) or getattr(f, "__module__", "") != vh.__name__:
        continue
    wrap(name)
step(); torch.cuda.synchronize()
tot = collections.defaultdict(lambda: [0, 0.0])
byname = collections.defaultdict(lambda: [0, 0.0])
for t, n, f in rows:
    tot[(n, f)][0] += 1; tot[(n, f)][1] += t
    byname[n][0] += 1; byname[n][1] += t
print(f"== {which}: {sum(t for t, _, _ in rows) / 1e3:.2f} ms in {len(rows)} calls (each call synchronised: launch latency included)")
for n, (c, t) in sorted(byname.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.0f} us x{c:4d}  {n}")
print("== by (entry point, arguments)")
for (n, f), (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t:8.0f} us x{c:2d}  {n} {f}")
