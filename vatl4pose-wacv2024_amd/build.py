"""Build libvatl_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python vatl4pose-wacv2024_amd/build.py [--force] [--ablation]

``--ablation`` builds the PROFILING variant libvatl_hip_ablation.so (-DVATL_ABLATION: schedule variants 10..13 and the
"no epilogue" / "one k-tile" knobs of tools/conv_bench.py, which compute wrong results by construction).  The product
library never contains them; the tools load the variant through VATL_HIP_LIB.

Objects go to vatl4pose-wacv2024_amd/build/, the library to
vatl4pose-wacv2024_amd/vatl_hip/libvatl_hip.so (in-tree, git-ignored: it
travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "vatl_hip", "libvatl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True, ablation: bool = False) -> str:
    obj_dir = OBJ + ("_ablation" if ablation else "")
    lib = LIB.replace("libvatl_hip.so", "libvatl_hip_ablation.so") if ablation else LIB
    flags = FLAGS + (["-DVATL_ABLATION"] if ablation else [])
    os.makedirs(obj_dir, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "vatl_hip.h")]
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s[:-4] + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not _newer(lib, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, ablation="--ablation" in sys.argv))
