"""Helpers shared by the -m gpu parity tests."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def to_dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).to(dev())


def record(name, **kw):
    """Append one line of parity evidence (kept under gpurun_out/, scratch)."""
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kw.items()}}) + "\n")
    except OSError:
        pass


def rel_err(got, ref):
    """max |got-ref| / max |ref|  (the 1e-4 'relative fp32' bar of BASELINE.json is read this way
    for dense tensors; element-wise checks are added where values are O(1))."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
