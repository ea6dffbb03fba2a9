"""Box format helpers used on the hot path (reference: alphapose/utils/bbox.py:74-105, 197-226)."""
import numpy as np


def bbox_xyxy_to_xywh(xyxy):
    """(xmin, ymin, xmax, ymax) -> (x, y, w, h) with the inclusive-pixel +1."""
    if isinstance(xyxy, (tuple, list)):
        if len(xyxy) != 4:
            raise IndexError(f"Bounding boxes must have 4 elements, given {len(xyxy)}")
        return (xyxy[0], xyxy[1], xyxy[2] - xyxy[0] + 1, xyxy[3] - xyxy[1] + 1)
    if isinstance(xyxy, np.ndarray):
        if xyxy.size % 4 != 0:
            raise IndexError(f"Bounding boxes must have n * 4 elements, given {xyxy.shape}")
        return np.hstack((xyxy[:, :2], xyxy[:, 2:4] - xyxy[:, :2] + 1))
    raise TypeError(f"Expect input xywh a list, tuple or numpy.ndarray, given {type(xyxy)}")


def _box_to_center_scale(x, y, w, h, aspect_ratio=1.0, scale_mult=1.25):
    """Box -> (center (2,) float32, scale (2,) float32) with the aspect fix-up and the 1.25 margin (bbox.py:197-214)."""
    center = np.zeros((2), dtype=np.float32)
    center[0] = x + w * 0.5
    center[1] = y + h * 0.5
    if w > aspect_ratio * h:
        h = w / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    scale = np.array([w * 1.0, h * 1.0], dtype=np.float32)
    if center[0] != -1:
        scale = scale * scale_mult
    return center, scale


def box_to_center_scale_batch(boxes_xyxy, aspect_ratio, scale_mult=1.25):
    """``_box_to_center_scale`` over (B,4) [xmin,ymin,xmax,ymax] float64 boxes -> centers, scales (B,2) float32."""
    b = np.asarray(boxes_xyxy, np.float64).reshape(-1, 4)
    x, y = b[:, 0], b[:, 1]
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    center = np.stack([x + w * 0.5, y + h * 0.5], 1).astype(np.float32)
    wide, tall = w > aspect_ratio * h, w < aspect_ratio * h
    h2 = np.where(wide, w / aspect_ratio, h)
    w2 = np.where(tall, h * aspect_ratio, w)
    scale = np.stack([w2, h2], 1).astype(np.float32)
    scale = np.where((center[:, :1] != -1), scale * np.float32(scale_mult), scale).astype(np.float32)
    return center, scale


def _center_scale_to_box(center, scale):
    """[xmin, ymin, xmax, ymax] of the crop window (bbox.py:217-226); float64 arithmetic on the float32 inputs, as
    numpy 1.23 (the reference's pin) promotes ``float32 scalar * python float``."""
    w, h = float(scale[0]) * 1.0, float(scale[1]) * 1.0
    xmin, ymin = float(center[0]) - w * 0.5, float(center[1]) - h * 0.5
    return [xmin, ymin, xmin + w, ymin + h]


def center_scale_to_box_batch(centers, scales):
    c, s = np.asarray(centers, np.float64), np.asarray(scales, np.float64)
    lo = c - s * 0.5
    return np.concatenate([lo, lo + s], 1)


def bbox_xywh_to_xyxy(xywh):
    """(x, y, w, h) -> (xmin, ymin, xmax, ymax) with the inclusive-pixel -1 (bbox.py:40-71)."""
    if isinstance(xywh, (tuple, list)):
        if len(xywh) != 4:
            raise IndexError(f"Bounding boxes must have 4 elements, given {len(xywh)}")
        w, h = np.maximum(xywh[2] - 1, 0), np.maximum(xywh[3] - 1, 0)
        return (xywh[0], xywh[1], xywh[0] + w, xywh[1] + h)
    if isinstance(xywh, np.ndarray):
        if xywh.size % 4 != 0:
            raise IndexError(f"Bounding boxes must have n * 4 elements, given {xywh.shape}")
        return np.hstack((xywh[:, :2], xywh[:, :2] + np.maximum(0, xywh[:, 2:4] - 1)))
    raise TypeError(f"Expect input xywh a list, tuple or numpy.ndarray, given {type(xywh)}")


def bbox_clip_xyxy(xyxy, width, height):
    """Clip (xmin, ymin, xmax, ymax) to the image (0, 0, width - 1, height - 1) (bbox.py:108-150)."""
    if isinstance(xyxy, (tuple, list)):
        if len(xyxy) != 4:
            raise IndexError(f"Bounding boxes must have 4 elements, given {len(xyxy)}")
        return (np.minimum(width - 1, np.maximum(0, xyxy[0])), np.minimum(height - 1, np.maximum(0, xyxy[1])),
                np.minimum(width - 1, np.maximum(0, xyxy[2])), np.minimum(height - 1, np.maximum(0, xyxy[3])))
    if isinstance(xyxy, np.ndarray):
        if xyxy.size % 4 != 0:
            raise IndexError(f"Bounding boxes must have n * 4 elements, given {xyxy.shape}")
        lim = np.array([width - 1, height - 1, width - 1, height - 1])
        return np.minimum(lim, np.maximum(0, xyxy))
    raise TypeError(f"Expect input xywh a list, tuple or numpy.ndarray, given {type(xyxy)}")
