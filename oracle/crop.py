"""numpy restatement of the reference's crop producer (TEST ORACLE).

The reference cuts one person crop per item with
``SimpleTransform.test_transform`` / ``__call__``
(alphapose/utils/presets/simple_transform.py:81-98, 179-251):

    bbox -> _box_to_center_scale (alphapose/utils/bbox.py:197-214)
         -> get_affine_transform (alphapose/utils/transforms.py:753-786, get_dir :313-321, get_3rd_point :307-310)
         -> cv2.warpAffine(img, trans, (w, h), flags=cv2.INTER_LINEAR)
         -> im_to_torch (transforms.py:76-91): HWC u8 -> CHW float, ``/255`` when max > 1
         -> ``img[c] += (-0.406, -0.457, -0.480)[c]``

PARITY UNPINNED for ``warp_affine_u8``: the arithmetic lives in a third-party dependency that is absent from
/root/reference and from this image (``opencv-python>=4.8.1.78``, pyproject.toml:67; ``import cv2`` fails here), and the
reference holds no fixture for it.  What follows restates the published OpenCV 4.8 algorithm
(modules/imgproc/src/imgwarp.cpp: ``cv::warpAffine``, ``WarpAffineInvoker``, ``initInterTab2D``, ``remapBilinear<FixedPtCast
<int, uchar, 15>, …>``): forward matrix inverted in double, 10-bit fixed-point coordinates with a 5-bit sub-pixel fraction,
int16 weights scaled by 2^15, BORDER_CONSTANT 0.  OpenCV >= 4.11 ships a second, float-coordinate code path for the same
call; results differ from the classic path by <= 1 grey level on a few pixels.  The pure numpy/torch pieces
(``box_to_center_scale``, ``affine_transform_matrix``, ``center_scale_to_box``, ``image_to_tensor``) ARE pinned: tests/golden/crop.npz holds
outputs of the reference's own functions (tools/make_golden.py::gen_crop).
"""
from __future__ import annotations

import numpy as np

MEAN = (0.406, 0.457, 0.480)          # simple_transform.py:93-95, 246-248

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
COEF_BITS = 15
COEF_SCALE = 1 << COEF_BITS


# --------------------------------------------------------------------------
# bbox -> centre / scale -> 2x3 matrix                      (pinned by fixtures)
# --------------------------------------------------------------------------
def box_to_center_scale(x, y, w, h, aspect_ratio=1.0, scale_mult=1.25):
    """bbox.py:197-214.  ``center``/``scale`` are float32 arrays; the aspect fix-up runs on the caller's scalars."""
    center = np.zeros(2, np.float32)
    center[0] = x + w * 0.5
    center[1] = y + h * 0.5
    if w > aspect_ratio * h:
        h = w / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    scale = np.array([w * 1.0, h * 1.0], np.float32)
    if center[0] != -1:
        scale = scale * scale_mult
    return center, scale


def center_scale_to_box(center, scale):
    """bbox.py:217-226 -> [xmin, ymin, xmax, ymax].

    The reference stack (numpy 1.23.5) promotes ``np.float32 scalar * python float`` to float64, so the arithmetic below
    is float64 on float32 inputs; the caller stores the result as float32 (``torch.tensor(bbox)``)."""
    w = float(scale[0]) * 1.0
    h = float(scale[1]) * 1.0
    xmin = float(center[0]) - w * 0.5
    ymin = float(center[1]) - h * 0.5
    return [xmin, ymin, xmin + w, ymin + h]


def affine_transform_matrix(center, scale, rot, output_size, inv=False) -> np.ndarray:
    """transforms.py:753-786: three point pairs in float32, 2x3 solution in float64 (cv2.getAffineTransform)."""
    center = np.asarray(center, np.float32)
    scale = np.asarray(scale, np.float32)
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    p = (0.0, float(src_w) * -0.5)
    src_dir = np.array([p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs], np.float64)         # get_dir :313-321
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = center.astype(np.float64) + src_dir                                           # float64 sum, float32 store
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    for pts in (src, dst):                                                                  # get_3rd_point :307-310
        d = pts[0] - pts[1]
        pts[2] = pts[1] + np.array([-d[1], d[0]], np.float32)
    a, b = (dst, src) if inv else (src, dst)
    lhs = np.concatenate([a.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(lhs, b.astype(np.float64)).T                                     # (2,3)


def transform_point(pt, t):
    """transforms.py:789-792 ``affine_transform``."""
    return (np.asarray(t, np.float64) @ np.array([pt[0], pt[1], 1.0]))[:2]


# --------------------------------------------------------------------------
# cv2.warpAffine, INTER_LINEAR, BORDER_CONSTANT(0), uint8          (unpinned)
# --------------------------------------------------------------------------
def invert_affine(m) -> np.ndarray:
    """cv::warpAffine without WARP_INVERSE_MAP inverts the 2x3 forward matrix in double (imgwarp.cpp)."""
    m = np.array(m, np.float64).reshape(6).copy()
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0] = a11
    m[1] *= -d
    m[3] *= -d
    m[4] = a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m.reshape(2, 3)


def bilinear_weight_table() -> np.ndarray:
    """``initInterTab2D(INTER_LINEAR, fixpt=true)``: (32*32, 4) int16 weights [w00, w01, w10, w11], index fy*32+fx.

    Every entry is exact ((32-fx)(32-fy)*32 …) except fx=fy=0, where 1.0*32768 saturates to 32767 and the table's sum
    correction adds the missing 1 to the diagonal neighbour: {32767, 0, 0, 1}."""
    t = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 4), np.int64)
    for fy in range(INTER_TAB_SIZE):
        for fx in range(INTER_TAB_SIZE):
            wy = (np.float32(1) - np.float32(fy) / np.float32(INTER_TAB_SIZE), np.float32(fy) / np.float32(INTER_TAB_SIZE))
            wx = (np.float32(1) - np.float32(fx) / np.float32(INTER_TAB_SIZE), np.float32(fx) / np.float32(INTER_TAB_SIZE))
            w = [int(np.clip(np.rint(np.float32(vy * vx) * np.float32(COEF_SCALE)), -32768, 32767)) for vy in wy for vx in wx]
            diff = sum(w) - COEF_SCALE
            if diff != 0:                       # only (0,0): the scan of the reference finds the last element as "largest"
                w[3] -= diff
            t[fy * INTER_TAB_SIZE + fx] = w
    return t.astype(np.int16)


_WTAB = None


def warp_affine_u8(src: np.ndarray, m, dsize) -> np.ndarray:
    """``cv2.warpAffine(src, m, dsize, flags=cv2.INTER_LINEAR)`` for HxWxC uint8, border constant 0.  dsize = (w, h)."""
    global _WTAB
    if _WTAB is None:
        _WTAB = bilinear_weight_table().astype(np.int64)
    src = np.ascontiguousarray(src)
    assert src.dtype == np.uint8 and src.ndim == 3
    sh, sw, cn = src.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    mi = invert_affine(m).reshape(6)
    rd = AB_SCALE // INTER_TAB_SIZE // 2
    xs = np.arange(dw, dtype=np.float64)
    ys = np.arange(dh, dtype=np.float64)
    adelta = np.rint(mi[0] * xs * AB_SCALE).astype(np.int64)              # saturate_cast<int>(double) = round half even
    bdelta = np.rint(mi[3] * xs * AB_SCALE).astype(np.int64)
    x0 = np.rint((mi[1] * ys + mi[2]) * AB_SCALE).astype(np.int64) + rd
    y0 = np.rint((mi[4] * ys + mi[5]) * AB_SCALE).astype(np.int64) + rd
    X = (x0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    w = _WTAB[(Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))]     # (dh,dw,4)
    s64 = src.astype(np.int64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
        v = s64[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        return np.where(ok[..., None], v, 0)

    acc = (tap(sy, sx) * w[..., 0:1] + tap(sy, sx + 1) * w[..., 1:2] + tap(sy + 1, sx) * w[..., 2:3] + tap(sy + 1, sx + 1) * w[..., 3:4])
    out = (acc + (1 << (COEF_BITS - 1))) >> COEF_BITS
    return np.clip(out, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------
# u8 crop -> normalised tensor                                        (pinned)
# --------------------------------------------------------------------------
def image_to_tensor(img_u8: np.ndarray) -> np.ndarray:
    """im_to_torch (transforms.py:76-91) + mean shift: (H,W,3) u8 -> (3,H,W) float32.

    The ``/255`` only happens when the crop's maximum exceeds 1 (an all-dark crop keeps raw 0/1 values)."""
    x = np.transpose(img_u8, (2, 0, 1)).astype(np.float32)
    if x.max() > 1:
        x = x / np.float32(255)
    for c in range(3):
        x[c] = x[c] + np.float32(-MEAN[c])
    return x


def test_transform(img_u8: np.ndarray, bbox, input_size=(256, 192)):
    """simple_transform.py:81-98 -> ((3,H,W) float32 crop, [xmin,ymin,xmax,ymax] of the aspect-corrected 1.25x box)."""
    xmin, ymin, xmax, ymax = bbox
    inp_h, inp_w = input_size
    center, scale = box_to_center_scale(xmin, ymin, xmax - xmin, ymax - ymin, float(inp_w) / inp_h)
    scale = scale * 1.0
    trans = affine_transform_matrix(center, scale, 0, [inp_w, inp_h])
    img = warp_affine_u8(img_u8, trans, (int(inp_w), int(inp_h)))
    return image_to_tensor(img), center_scale_to_box(center, scale)


def float_bilinear_reference(src: np.ndarray, m, dsize) -> np.ndarray:
    """Independent float64 bilinear warp (no fixed point): the sanity bound for ``warp_affine_u8`` in the tests."""
    sh, sw, _ = src.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    mi = invert_affine(m)
    gx, gy = np.meshgrid(np.arange(dw, dtype=np.float64), np.arange(dh, dtype=np.float64))
    fx = mi[0, 0] * gx + mi[0, 1] * gy + mi[0, 2]
    fy = mi[1, 0] * gx + mi[1, 1] * gy + mi[1, 2]
    x0 = np.floor(fx).astype(np.int64)
    y0 = np.floor(fy).astype(np.int64)
    ax, ay = (fx - x0)[..., None], (fy - y0)[..., None]
    s = src.astype(np.float64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
        return np.where(ok[..., None], s[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)], 0.0)

    return (tap(y0, x0) * (1 - ax) * (1 - ay) + tap(y0, x0 + 1) * ax * (1 - ay) + tap(y0 + 1, x0) * (1 - ax) * ay + tap(y0 + 1, x0 + 1) * ax * ay)
