import sys, os, time, cProfile, pstats
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")): sys.path.insert(0, p)
import torch
from bench import build_model
from alphapose.models import hip_engine
dev = torch.device("cuda:0")
m = build_model(dev)
x = torch.rand((4, 3, 256, 192), device=dev) - 0.45
def run(n):
    with torch.no_grad():
        for _ in range(n): m(x)
    torch.cuda.synchronize()
run(20)
for guard in (True, False, True, False):
    hip_engine.PARAM_GUARD = guard; hip_engine.invalidate(m); run(20)
    t0 = time.perf_counter(); run(300); dt = (time.perf_counter() - t0) / 300
    print("guard", guard, "ms/call", round(dt * 1e3, 4), flush=True)
for iv in (1e9, 0.02, 0.005, 0.0):
    hip_engine.PARAM_GUARD = True; hip_engine.GUARD_MIN_INTERVAL_S = iv; hip_engine.invalidate(m); run(20)
    t0 = time.perf_counter(); run(300); dt = (time.perf_counter() - t0) / 300
    print("guard on, interval", iv, "ms/call", round(dt * 1e3, 4), flush=True)
hip_engine.GUARD_MIN_INTERVAL_S = 0.005
hip_engine.PARAM_GUARD = True; hip_engine.invalidate(m); run(20)
pr = cProfile.Profile(); pr.enable(); run(300); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
g = m.__dict__["_vatl_plan"][2]
print("pending", len(g.pending))
