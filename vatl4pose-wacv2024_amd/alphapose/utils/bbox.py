"""Box format helpers used on the hot path (reference: alphapose/utils/bbox.py:74-105)."""
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
