"""Heat-map -> key-point decoders with the reference's call signatures, running on MI355X.

Reference: alphapose/utils/transforms.py — heatmap_to_coord_simple :550-583,
heatmap_to_coord_simple_regress :586-642, get_max_pred :710-727,
get_max_pred_batch :730-749, get_func_heatmap_to_coord :946-954.
The per-item functions keep their signatures and results (numpy in, numpy out) but
execute one libvatl_hip.so launch; the batched variants avoid the per-item D2H sync.
Crop / flip / SMPL helpers of the reference file are host-side data preparation and
out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

import vatl_hip as vh


def _dev_f32(a):
    if isinstance(a, torch.Tensor):
        t = a.detach()
    else:
        t = torch.as_tensor(np.ascontiguousarray(a))
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise vh.VatlError("heat-map decoding runs on MI355X only (no CPU fallback)")
        t = t.cuda()
    return t.float().contiguous()


def heatmap_to_coord_batch(hms, bboxes):
    """(N,J,H,W) heat-maps + (N,4) xyxy boxes -> coords (N,J,2), maxvals (N,J,1), idx (N,J) device tensors."""
    coords, maxv, idx = vh.decode(_dev_f32(hms), _dev_f32(bboxes))
    return coords, maxv.unsqueeze(-1), idx


def heatmap_to_coord_simple(hms, bbox, hms_flip=None, **kwargs):
    """One item: (J,H,W) heat-maps, [xmin,ymin,xmax,ymax] -> (preds (J,2) f32, maxvals (J,1) f32) numpy."""
    h = _dev_f32(hms)
    if hms_flip is not None:
        h = (h + _dev_f32(hms_flip)) / 2
    b = _dev_f32(np.asarray(bbox, dtype=np.float32).reshape(1, 4)) if not isinstance(bbox, torch.Tensor) else _dev_f32(bbox).reshape(1, 4)
    coords, maxv, _ = vh.decode(h.unsqueeze(0), b)
    return coords[0].cpu().numpy(), maxv[0].unsqueeze(-1).cpu().numpy()


def heatmap_to_coord_simple_regress(preds, bbox, hm_shape, norm_type, hms_flip=None):
    """Soft-arg-max decode (LOSS.TYPE 'L1JointRegression'); preds (J,H,W) or (1,J,H,W)."""
    h = _dev_f32(preds)
    if h.dim() == 3:
        h = h.unsqueeze(0)
    b = _dev_f32(np.asarray(bbox, dtype=np.float32).reshape(1, 4)).expand(h.shape[0], 4).contiguous()
    coords, scores = vh.decode_softargmax(h, b, norm_type)
    if hms_flip is not None:
        f = _dev_f32(hms_flip)
        f = f.unsqueeze(0) if f.dim() == 3 else f
        c2, s2 = vh.decode_softargmax(f, b, norm_type)
        coords, scores = (coords + c2) / 2, (scores + s2) / 2
    coords, scores = coords.cpu().numpy(), scores.unsqueeze(-1).cpu().numpy()
    return (coords[0], scores[0]) if coords.shape[0] == 1 else (coords, scores)


def get_max_pred_batch(batch_heatmaps):
    """(B,J,H,W) -> preds (B,J,2) f32 heat-map pixel coords (zeroed where max <= 0), maxvals (B,J,1)."""
    h = _dev_f32(batch_heatmaps)
    _, maxv, idx = vh.decode(h, torch.zeros((h.shape[0], 4), device=h.device))
    w = h.shape[3]
    preds = torch.stack([(idx % w).float(), (idx // w).float()], dim=2) * (maxv > 0).unsqueeze(-1).float()
    return preds.cpu().numpy(), maxv.unsqueeze(-1).cpu().numpy()


def get_max_pred(heatmaps):
    p, m = get_max_pred_batch(_dev_f32(heatmaps).unsqueeze(0))
    return p[0], m[0]


def get_func_heatmap_to_coord(cfg):
    if cfg.DATA_PRESET.TYPE == "simple":
        if cfg.LOSS.TYPE == "MSELoss":
            return heatmap_to_coord_simple
        if cfg.LOSS.TYPE == "L1JointRegression":
            return heatmap_to_coord_simple_regress
        if cfg.LOSS.TYPE == "Combined":
            return [heatmap_to_coord_simple, heatmap_to_coord_simple_regress]
    raise NotImplementedError(f"no heat-map decoder for preset {cfg.DATA_PRESET.TYPE!r} / loss {cfg.LOSS.TYPE!r}")


def flip(x):
    """Horizontal flip of an NCHW / CHW tensor (test-time augmentation helper, transforms.py:479-483)."""
    assert x.dim() in (3, 4)
    return x.flip(dims=(x.dim() - 1,))


def flip_heatmap(heatmap, joint_pairs, shift=False):
    """Flip heat-maps and swap left/right joints (transforms.py:486-518)."""
    assert heatmap.dim() in (3, 4)
    out = flip(heatmap)
    ndim = out.dim()
    if ndim == 3:
        out = out.unsqueeze(0)
    for a, b in joint_pairs:
        idx = torch.tensor((b, a), device=out.device)
        inv = torch.tensor((a, b), device=out.device)
        out[:, idx] = out[:, inv]
    if shift:
        out[:, :, :, 1:] = out[:, :, :, 0:-1].clone()
    return out.squeeze(0) if ndim == 3 else out
