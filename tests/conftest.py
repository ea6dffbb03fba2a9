import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vatl4pose-wacv2024_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _granted_cpus():
    """CPUs the container may USE (cgroup CPU-time quota), not the ones it shows: the GPU boxes show 256 and grant 16."""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return visible if q == "max" else max(1, min(visible, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        return visible


# The oracle's CPU forwards (float64 networks among them) run on torch's intra-op pool: one worker per VISIBLE core by default, which on a box that grants 16 of 256
# gets the whole test process throttled (a 32-crop fp32 forward: 36.8 s on 256 threads, 1.1 s on 16 — profiles/r06_notes.md).  Sized before torch / numpy start
# their pools; child processes (bench.py, torch.distributed.run) inherit the setting.
for _v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_v, str(_granted_cpus()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_scorers():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "scorers.npz"))


@pytest.fixture(scope="session")
def golden_simplepose():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "simplepose_r50.npz"))
