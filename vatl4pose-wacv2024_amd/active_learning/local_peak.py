"""Local-peak statistics of heat-maps on MI355X (reference: active_learning/local_peak.py:5-22).

A pixel is a peak when it equals the maximum of its 3x3 neighbourhood with a constant-0
border (scipy ``maximum_filter(mode='constant')``); peaks below ``order`` x the largest
peak are dropped.  ``localpeak_mean`` pools the kept peaks of all joints.
"""
from __future__ import annotations

import numpy as np
import torch

import vatl_hip as vh


def _dev(a):
    t = a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a))
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise vh.VatlError("local-peak scoring runs on MI355X only (no CPU fallback)")
        t = t.cuda()
    return t.float().contiguous()


def localpeak_values(image, filter_size=3, order=0.5):
    """Kept peak values of one (H,W) map in row-major order (numpy array)."""
    if filter_size != 3:
        raise NotImplementedError("only the reference's 3x3 footprint is implemented")
    t = _dev(image)
    return t[vh.localpeak_mask(t.unsqueeze(0), order)[0].bool()].cpu().numpy()


def localpeak_mean(heatmaps, filter_size=3, order=0.5):
    """Mean of the kept peaks of all joints of one item, heat-maps (J,H,W); nan when none."""
    if filter_size != 3:
        raise NotImplementedError("only the reference's 3x3 footprint is implemented")
    mean, _ = vh.localpeak_mean(_dev(heatmaps).unsqueeze(0), order)
    return np.float32(mean.item())


def localpeak_mean_batch(heatmaps, order=0.5):
    """(N,J,H,W) device tensor -> (N,) means, (N,J) kept-peak counts (device tensors)."""
    return vh.localpeak_mean(_dev(heatmaps), order)
