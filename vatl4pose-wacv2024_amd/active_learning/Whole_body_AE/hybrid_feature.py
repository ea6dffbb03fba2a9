"""Hand-crafted whole-body pose feature (reference: Whole_body_AE/hybrid_feature.py:6-59).

42 values for 17 key-points: score-weighted-centroid-relative x and y divided by the box
height, plus 8 joint-triangle angles ``atan(|(m1-m2)/(1+m1*m2+eps)|)``.  float64 in,
float64 out, computed by ``vatl_hybrid_feature_f64``; the fused scoring path
(``vatl_hybrid_ae_wpu``) evaluates the same formulas in-kernel.
"""
from __future__ import annotations

import numpy as np
import torch

import vatl_hip as vh


def compute_hybrid_batch(bboxes_xywh, keypoints):
    """(N,4) boxes (x,y,w,h), (N,51) key-points -> (N,42) float64 device tensor, (N,) status."""
    dev = torch.device("cuda", torch.cuda.current_device())
    b = torch.as_tensor(np.asarray(bboxes_xywh, np.float64)).reshape(-1, 4).to(dev).contiguous()
    k = torch.as_tensor(np.asarray(keypoints, np.float64)).reshape(-1, 51).to(dev).contiguous()
    return vh.hybrid_feature_f64(k, b)


def compute_hybrid(bbox, keypoints):
    """bbox [x,y,w,h], keypoints 51 values (x,y,score)*17 -> ndarray (42,) float64."""
    if not torch.cuda.is_available():
        raise vh.VatlError("compute_hybrid runs on MI355X only (no CPU fallback)")
    feat, status = compute_hybrid_batch([bbox], [keypoints])
    st = int(status.item())
    assert st != 1, "height of human body must be positive!"
    assert st != 2, "at least one visible keypoint is required!"
    return feat[0].cpu().numpy()
