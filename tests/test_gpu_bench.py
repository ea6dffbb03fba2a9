"""bench.py keeps its contract: one JSON line with the metric, the roofline and (at N = 1) the CPU baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.5 < r["frac"] < 1.0 and d["value"] > 5000
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # whole-job rate is consistent with the step time
    assert abs(d["value"] - 1024 * 1000.0 / d["ms_per_step"]) / d["value"] < 0.01
