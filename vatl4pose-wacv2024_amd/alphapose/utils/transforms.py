"""Heat-map -> key-point decoders with the reference's call signatures, running on MI355X.

Reference: alphapose/utils/transforms.py — heatmap_to_coord_simple :550-583,
heatmap_to_coord_simple_regress :586-642, get_max_pred :710-727,
get_max_pred_batch :730-749, get_func_heatmap_to_coord :946-954.
The per-item functions keep their signatures and results (numpy in, numpy out) but
execute one libvatl_hip.so launch; the batched variants avoid the per-item D2H sync.
The crop geometry of the same file (get_affine_transform :753-786, affine_transform
:789-792, flip_joints_3d :521-547, im_to_torch :76-91) is kept as host numpy, batched:
it is O(crops) float64 work whose result (one 2x3 matrix per crop) feeds the warp
kernel (``vatl_hip.crop_warp_affine``).  SMPL helpers are out of
scope.
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
    return get_max_pred_batch_begin(batch_heatmaps)()


def get_max_pred_batch_begin(batch_heatmaps):
    """``get_max_pred_batch`` with the read-back deferred: the decode is enqueued now, the returned callable copies the results to the host."""
    h = _dev_f32(batch_heatmaps)
    _, maxv, idx = vh.decode(h, torch.zeros((h.shape[0], 4), device=h.device))
    w = h.shape[3]
    preds = torch.stack([(idx % w).float(), (idx // w).float()], dim=2) * (maxv > 0).unsqueeze(-1).float()
    return lambda: (preds.cpu().numpy(), maxv.unsqueeze(-1).cpu().numpy())


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


# ---------------------------------------------------------------------------------------------
# crop geometry (host, float64 like the reference's numpy / cv2.getAffineTransform arithmetic)
# ---------------------------------------------------------------------------------------------
def get_affine_transform_batch(centers, scales, rots, output_size, inv=False):
    """transforms.py:753-786 for B crops at once: centers/scales (B,2) float32, rots (B,) degrees, output_size
    [w, h] -> (B,2,3) float64.  The three point pairs are rounded to float32 exactly where the reference stores
    them in float32 arrays; the 3-point system is then solved in float64 (cv2.getAffineTransform)."""
    c = np.asarray(centers, np.float32).reshape(-1, 2)
    sc = np.asarray(scales, np.float32).reshape(-1, 2)
    n = c.shape[0]
    rad = np.pi * np.broadcast_to(np.asarray(rots, np.float64), (n,)) / 180
    sn, cs = np.sin(rad), np.cos(rad)
    half = sc[:, 0].astype(np.float64) * -0.5
    src_dir = np.stack([0.0 * cs - half * sn, 0.0 * sn + half * cs], 1)
    dst_w, dst_h = output_size[0], output_size[1]
    src = np.zeros((n, 3, 2), np.float32)
    dst = np.zeros((n, 3, 2), np.float32)
    src[:, 0] = c
    src[:, 1] = c.astype(np.float64) + src_dir
    dst[:, 0] = [dst_w * 0.5, dst_h * 0.5]
    dst[:, 1] = [dst_w * 0.5, dst_h * 0.5 + dst_w * -0.5]
    for pts in (src, dst):
        d = pts[:, 0] - pts[:, 1]
        pts[:, 2] = pts[:, 1] + np.stack([-d[:, 1], d[:, 0]], 1)
    a, b = (dst, src) if inv else (src, dst)
    lhs = np.concatenate([a.astype(np.float64), np.ones((n, 3, 1))], axis=2)
    return np.transpose(np.linalg.solve(lhs, b.astype(np.float64)), (0, 2, 1))


def get_affine_transform(center, scale, rot, output_size, shift=None, inv=0):
    """Per-item form with the reference's signature (transforms.py:753-786); ``shift`` other than zero is unused there."""
    if shift is not None and np.any(np.asarray(shift) != 0):
        raise NotImplementedError("get_affine_transform: non-zero shift is not used by any caller of the reference")
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale])
    return get_affine_transform_batch(np.asarray(center)[None], np.asarray(scale)[None], rot, output_size, bool(inv))[0]


def affine_transform(pt, t):
    """transforms.py:789-792."""
    return np.dot(t, np.array([pt[0], pt[1], 1.0]).T)[:2]


def invert_affine_batch(m):
    """(B,2,3) forward matrices -> the dst->src maps cv::warpAffine derives from them (same float64 operation order)."""
    m = np.array(m, np.float64).reshape(-1, 6)
    d = m[:, 0] * m[:, 4] - m[:, 1] * m[:, 3]
    d = np.where(d != 0, 1.0 / np.where(d != 0, d, 1.0), 0.0)
    out = np.empty_like(m)
    out[:, 0], out[:, 4] = m[:, 4] * d, m[:, 0] * d
    out[:, 1], out[:, 3] = m[:, 1] * -d, m[:, 3] * -d
    out[:, 2] = -out[:, 0] * m[:, 2] - out[:, 1] * m[:, 5]
    out[:, 5] = -out[:, 3] * m[:, 2] - out[:, 4] * m[:, 5]
    return out.reshape(-1, 2, 3)


def flip_joints_3d(joints_3d, width, joint_pairs):
    """(J,3,2) joints mirrored about the image's vertical axis with left/right swapped (transforms.py:521-547)."""
    joints = joints_3d.copy()
    joints[:, 0, 0] = width - joints[:, 0, 0] - 1
    for a, b in joint_pairs:
        joints[[a, b]] = joints[[b, a]]
    joints[:, :, 0] *= joints[:, :, 1]
    return joints


def im_to_torch(img):
    """(H,W,3) ndarray -> (3,H,W) float tensor, divided by 255 when its maximum exceeds 1 (transforms.py:76-91).
    Layout helper for callers that hold a host image; the crop path fuses this into the warp kernel."""
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(img, (2, 0, 1)))).float()
    if t.max() > 1:
        t /= 255
    return t
