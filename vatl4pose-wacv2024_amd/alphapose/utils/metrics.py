"""Running averages and heat-map accuracy (reference: alphapose/utils/metrics.py).

``DataLogger`` :14-32 and ``calc_accuracy`` :118-147 (+ calc_dist :221-236, dist_acc
:239-245) run inside the fine-tune loop (ActiveLearning.py:670-676).  The arg-max of
predictions and labels comes from the same HIP kernel as the decoder, so only
(B,J) indices cross to the host instead of two (B,J,H,W) tensors.
``evaluate_mAP`` (:65-115) drives the third-party COCO API when it is installed (offline evaluation, not re-implemented).
"""
from __future__ import annotations

import numpy as np

from .transforms import get_max_pred_batch


class DataLogger:
    """Average data logger."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.value = 0
        self.sum = 0
        self.cnt = 0
        self.avg = 0

    def update(self, value, n=1):
        self.value = value
        self.sum += value * n
        self.cnt += n
        self._cal_avg()

    def _cal_avg(self):
        self.avg = self.sum / self.cnt


def calc_accuracy(preds, labels, thr=0.5):
    """Fraction of joints whose heat-map arg-max lies within ``thr`` of the label's
    arg-max, distances normalised by (W,H)/10, averaged over joints that have at
    least one labelled item (label arg-max > 1 in both axes)."""
    return calc_accuracy_begin(preds, labels, thr)()


def calc_accuracy_begin(preds, labels, thr=0.5):
    """``calc_accuracy`` in two halves: the arg-max decodes are enqueued now (device work), the returned callable reads them back and does the
    host arithmetic — so a training loop can log the accuracy of a step without waiting for the step (the read-back of a (B, J, 2) tensor
    drains the stream; the reference's per-iteration ``loss.item()`` does the same, but there the loader runs in worker processes)."""
    from .transforms import get_max_pred_batch_begin
    hm_h, hm_w = preds.shape[2], preds.shape[3]
    p_fin, t_fin = get_max_pred_batch_begin(preds), get_max_pred_batch_begin(labels)

    def finish():
        p, _ = p_fin()
        t, _ = t_fin()
        norm = np.array([hm_w, hm_h], np.float64) / 10
        valid = (t[..., 0] > 1) & (t[..., 1] > 1)                       # (B,J)
        d = np.linalg.norm(p.astype(np.float32) / norm - t.astype(np.float32) / norm, axis=2)
        d = np.where(valid, d, 0.0).T                                   # (J,B); 0 marks "not counted" like the reference
        total, cnt = 0.0, 0
        for row in d:
            used = row != 0
            if used.sum() > 0:
                total += float((row[used] < thr).sum()) / used.sum()
                cnt += 1
        return total / cnt if cnt > 0 else 0
    return finish


def have_coco_tools() -> bool:
    """Whether evaluate_mAP can run here (the COCO API is importable)."""
    import importlib.util
    return importlib.util.find_spec("pycocotools") is not None


def evaluate_mAP(res_file, ann_type="bbox", ann_file="./data/coco/annotations/person_keypoints_val2017.json", silence=False):
    """COCO evaluation of a result json against a ground-truth json (metrics.py:65-115) -> {'AP', 'AP .5', ..., 'AR'}.

    The arithmetic is the COCO API's (``pycocotools`` / AlphaPose's ``halpecocotools`` fork), a third-party package that is
    part of the reference's environment but not of this image; it is used when importable and never re-implemented here
    (offline evaluation).  Without it: NotImplementedError."""
    try:
        from pycocotools.coco import COCO
        from pycocotools.cocoeval import COCOeval
    except ImportError as e:
        raise NotImplementedError("evaluate_mAP needs the COCO API (pycocotools / halpecocotools), which is not installed; "
                                  "the result files are written for it (predicted_kpt.json / GT_kpt.json)") from e
    import contextlib
    import io
    sink = io.StringIO() if silence else None
    with (contextlib.redirect_stdout(sink) if silence else contextlib.nullcontext()):
        gt = COCO(ann_file)
        ev = COCOeval(gt, gt.loadRes(res_file), ann_type)
        ev.evaluate()
        ev.accumulate()
        ev.summarize()
    if isinstance(ev.stats[0], dict):                               # Halpe full-body fork: per-part dictionaries
        return {part: ev.stats[i][part][0] for i, part in enumerate(["body", "foot", "face", "hand", "fullbody"])}
    names = ["AP", "AP .5", "AP .6", "AP .7", "AP .75", "AP .8", "AP .95", "AP (M)", "AP (L)", "AR"]
    return {name: ev.stats[i] for i, name in enumerate(names)}
