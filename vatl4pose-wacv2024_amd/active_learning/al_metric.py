"""Active-learning metrics on the hot loop (reference: active_learning/al_metric.py).

``compute_OKS`` (:42-69) runs once per item in the reference; ``compute_OKS_batch`` evaluates a whole batch on
the host in one vectorised numpy call (§8f rank 1: needs ground truth, only meaningful with real data).
``compute_alc`` (:31-36) is the area under the learning curve.  Plot helpers are out of scope.
"""
from __future__ import annotations

import numpy as np

OKS_sigmas = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0
OKS_vars = (OKS_sigmas * 2) ** 2
OKS_k = len(OKS_sigmas)


def compute_OKS_batch(bboxes_xywh, pred, gt):
    """bboxes (N,4) xywh, pred/gt (N,51) -> (N,) OKS with the box area as the object scale."""
    b = np.asarray(bboxes_xywh, np.float64).reshape(-1, 4)
    d = np.asarray(pred, np.float64).reshape(-1, 17, 3)
    g = np.asarray(gt, np.float64).reshape(-1, 17, 3)
    vis = g[:, :, 2] > 0
    any_vis = vis.any(axis=1)
    dx, dy = d[:, :, 0] - g[:, :, 0], d[:, :, 1] - g[:, :, 1]
    z = np.zeros_like(dx)
    x0, x1 = (b[:, 0] - b[:, 2])[:, None], (b[:, 0] + 2 * b[:, 2])[:, None]
    y0, y1 = (b[:, 1] - b[:, 3])[:, None], (b[:, 1] + 2 * b[:, 3])[:, None]
    dx_far = np.maximum(z, x0 - d[:, :, 0]) + np.maximum(z, d[:, :, 0] - x1)
    dy_far = np.maximum(z, y0 - d[:, :, 1]) + np.maximum(z, d[:, :, 1] - y1)
    dx = np.where(any_vis[:, None], dx, dx_far)
    dy = np.where(any_vis[:, None], dy, dy_far)
    e = (dx ** 2 + dy ** 2) / OKS_vars / ((b[:, 2] * b[:, 3])[:, None] + np.spacing(1)) * 0.5
    use = np.where(any_vis[:, None], vis, True)
    return (np.exp(-e) * use).sum(axis=1) / use.sum(axis=1)


def compute_OKS(bb, predkpts, GTkpts):
    return float(compute_OKS_batch([bb], [predkpts], [GTkpts])[0])


def compute_alc(percentages, performances):
    x, y = 0.01 * np.asarray(percentages, np.float64), 0.01 * np.asarray(performances, np.float64)
    return float(np.trapz(y, x))


def plot_learning_curves(*args, **kwargs):
    raise NotImplementedError("plotting is outside the MI355X hot path (SURVEY.md §2.1 row 13)")
