"""Batched uncertainty scoring of a stream of heat-maps (the fast path beside the
reference's per-item Python loop, ActiveLearning.py:299-414).

Inputs stay on the device; every scorer is one libvatl_hip.so launch over the
whole batch instead of per-item ``.cpu().numpy()`` round trips:

  key-points / HP   heatmap_to_coord_simple            ActiveLearning.py:304-306, 329-330
  THC               compute_thc + isPrev/isNext rule   :345-363, 747-760
  WPU               compute_hybrid -> AE -> MSELoss    :364-386
  local-peak mean   localpeak_mean ("combine weight")  :411-414
  MPE / Margin / Entropy   compute_mpe / compute_margin / compute_entropy   :387-396, 762-796 (``multi_peak_scores``)

The stream is id-sorted and de-duplicated: the "prev"/"next" crops of item i are
the "current" crops of items i-1 / i+1 when isPrev / isNext hold (SURVEY.md §9
item 14), so one backbone forward per item serves all three heat-maps.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

import vatl_hip as vh


@dataclass
class Scores:
    keypoints: torch.Tensor          # (N,17,3) x, y, score  — fp32, image pixels
    argmax: torch.Tensor             # (N,17) int32 flat arg-max indices
    hp: torch.Tensor                 # (N,) -sum of joint scores ("HP" uncertainty)
    pose_score: torch.Tensor         # (N,) float64: mean + 1.25 max of joint scores (json "score", numpy 1.23 promotion)
    localpeak: torch.Tensor          # (N,) mean of kept local peaks (nan when none)
    thc: torch.Tensor | None = None  # (N,)
    wpu: torch.Tensor | None = None  # (N,)
    wpu_status: torch.Tensor | None = None


def score_batch(heatmaps: torch.Tensor, bboxes: torch.Tensor, is_prev: torch.Tensor | None = None,
                is_next: torch.Tensor | None = None, thc_norm: str | None = "L1", ae_flat: torch.Tensor | None = None,
                ae_dims=(42, 4), wpu_only38: bool = False) -> Scores:
    """heatmaps (N,J,H,W) fp32 on the device, bboxes (N,4) crop boxes xyxy."""
    kpts, idx, hp, pose_score = vh.decode_pose(heatmaps, bboxes)     # key-point rows + HP + json score: two launches, no torch glue kernels
    lp, _ = vh.localpeak_mean(heatmaps)
    out = Scores(keypoints=kpts, argmax=idx, hp=hp, pose_score=pose_score, localpeak=lp)
    if thc_norm is not None and is_prev is not None:
        out.thc = vh.thc_stream(heatmaps, is_prev.to(torch.uint8), is_next.to(torch.uint8), thc_norm)
    if ae_flat is not None:
        out.wpu, out.wpu_status = vh.hybrid_ae_wpu(kpts, bboxes, ae_flat, ae_dims[0], ae_dims[1], wpu_only38)
    return out


def multi_peak_scores(heatmaps: torch.Tensor, which: str) -> torch.Tensor:
    """(N,J,H,W) -> (N,) float64: the reference's 'MPE' / 'Margin' / 'Entropy' uncertainty of every item
    (ActiveLearning.py:762-796); per-plane terms from one launch, summed over the joints in float64 like the
    reference's Python accumulation."""
    if which == "Entropy":
        return vh.plane_entropy(heatmaps).double().sum(dim=1)
    _, _, _, mpe, margin = vh.peaks5(heatmaps, 5)
    if which == "MPE":
        return mpe.double().sum(dim=1)
    if which == "Margin":
        return margin.double().sum(dim=1)
    raise ValueError("Uncertainty type is not supported")
