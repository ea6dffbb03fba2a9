"""numpy restatement of the reference's per-item scoring formulas (TEST ORACLE).

Each function cites the reference lines it follows (paths relative to
/root/reference).  Everything is per item (one person crop) unless the name
says ``_batch``; heat-maps are float32 ``(J, H, W)`` arrays.

Pinned by tests/golden/scorers.npz (outputs of the reference itself, see
tools/make_golden.py).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# a8  hard arg-max decode                      alphapose/utils/transforms.py
# --------------------------------------------------------------------------

def argmax_peaks(hm: np.ndarray):
    """Flat first-max per joint.  Follows get_max_pred (transforms.py:710-727).

    Returns (idx int64 (J,), xy float32 (J,2), maxval float32 (J,1)).  ``xy`` is
    zeroed where ``maxval <= 0`` exactly as the reference's ``pred_mask`` does.
    """
    J, H, W = hm.shape
    flat = hm.reshape(J, H * W)
    idx = flat.argmax(axis=1)                       # first maximum, row-major
    maxval = flat[np.arange(J), idx].astype(np.float32).reshape(J, 1)
    xy = np.empty((J, 2), np.float32)
    xy[:, 0] = (idx % W).astype(np.float32)
    xy[:, 1] = (idx // W).astype(np.float32)
    xy *= (maxval > 0.0).astype(np.float32)
    return idx.astype(np.int64), xy, maxval


def quarter_pixel_shift(hm: np.ndarray, xy: np.ndarray) -> np.ndarray:
    """+-0.25 px refinement toward the larger neighbour (transforms.py:558-568).

    Applied only when ``1 < px < W-1`` and ``1 < py < H-1`` (strict), using the
    coordinates *after* the ``maxval <= 0`` zeroing.  Returns int8 (J,2) signs.
    """
    J, H, W = hm.shape
    sgn = np.zeros((J, 2), np.int8)
    for p in range(J):
        px = int(round(float(xy[p, 0])))
        py = int(round(float(xy[p, 1])))
        if 1 < px < W - 1 and 1 < py < H - 1:
            dx = hm[p, py, px + 1] - hm[p, py, px - 1]
            dy = hm[p, py + 1, px] - hm[p, py - 1, px]
            sgn[p, 0] = np.sign(dx)
            sgn[p, 1] = np.sign(dy)
    return sgn


def crop_to_image_affine(bbox, hm_w: int, hm_h: int) -> np.ndarray:
    """2x3 float64 map heat-map px -> image px for an un-rotated crop.

    Follows transform_preds -> get_affine_transform(inv=1) (transforms.py:
    704-708, 753-786): three point pairs are built in *float32* (centre,
    centre shifted up by half the box WIDTH, and their perpendicular third
    point), then the 2x3 transform is the exact solution of the 3-point system
    in float64 (cv2.getAffineTransform in the reference).  Only the box width
    enters the scale (transforms.py:763-769).
    """
    xmin, ymin, xmax, ymax = [float(v) for v in bbox]
    w = xmax - xmin
    h = ymax - ymin
    cx = xmin + w * 0.5
    cy = ymin + h * 0.5
    img = np.zeros((3, 2), np.float32)          # points in image space
    hmp = np.zeros((3, 2), np.float32)          # points in heat-map space
    img[0] = (cx, cy)
    img[1] = (cx + 0.0, cy + w * -0.5)
    hmp[0] = (hm_w * 0.5, hm_h * 0.5)
    hmp[1] = (hm_w * 0.5, hm_h * 0.5 + hm_w * -0.5)
    for pts in (img, hmp):                       # third point: rotate (p0-p1) by 90 deg about p1
        d = pts[0] - pts[1]
        pts[2] = pts[1] + np.array([-d[1], d[0]], np.float32)
    lhs = np.concatenate([hmp.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(lhs, img.astype(np.float64)).T      # (2,3)


def decode_heatmaps(hm: np.ndarray, bbox):
    """heatmap_to_coord_simple (transforms.py:550-583) for one item.

    Returns dict(idx int64 (J,), sign int8 (J,2), coords float32 (J,2) in image
    space, maxvals float32 (J,1)).
    """
    J, H, W = hm.shape
    idx, xy, maxval = argmax_peaks(hm)
    sgn = quarter_pixel_shift(hm, xy)
    xy = xy + sgn.astype(np.float32) * np.float32(0.25)
    T = crop_to_image_affine(bbox, W, H)
    out = np.zeros((J, 2), np.float32)
    for p in range(J):
        v = T @ np.array([float(xy[p, 0]), float(xy[p, 1]), 1.0])
        out[p] = v                                # float64 -> float32 store
    return dict(idx=idx, sign=sgn, coords=out, maxvals=maxval)


def decode_closed_form(hm: np.ndarray, bbox):
    """Same result as decode_heatmaps via the closed form used by the HIP kernel.

    X = cx32 + (u - W/2) * d / (W/2),  Y = cy32 + (v - H/2) * d / (W/2) with
    cx32/cy32 the float32-rounded box centre and d = cy32 - fl32(cy - w/2)
    (float32 subtraction), everything else float64.
    """
    J, H, W = hm.shape
    idx, xy, maxval = argmax_peaks(hm)
    sgn = quarter_pixel_shift(hm, xy)
    xy = xy + sgn.astype(np.float32) * np.float32(0.25)
    xmin, ymin, xmax, ymax = [float(v) for v in bbox]
    w = xmax - xmin
    h = ymax - ymin
    cx32 = np.float32(xmin + w * 0.5)
    cy32 = np.float32(ymin + h * 0.5)
    top32 = np.float32((ymin + h * 0.5) + w * -0.5)
    d = float(np.float32(cy32 - top32))
    g = d / (W * 0.5)
    out = np.zeros((J, 2), np.float32)
    out[:, 0] = (float(cx32) + (xy[:, 0].astype(np.float64) - W * 0.5) * g)
    out[:, 1] = (float(cy32) + (xy[:, 1].astype(np.float64) - H * 0.5) * g)
    return dict(idx=idx, sign=sgn, coords=out, maxvals=maxval)


def keypoints_51(coords: np.ndarray, maxvals: np.ndarray) -> np.ndarray:
    """(x, y, score) interleaved, ActiveLearning.py:305-306."""
    return np.concatenate([coords, maxvals], axis=1).reshape(-1)


def pose_score(maxvals: np.ndarray) -> float:
    """float(np.mean(s) + 1.25 * np.max(s)), ActiveLearning.py:314, as the reference's pinned numpy==1.23.5 evaluates it: np.mean of the float32
    scores is a float32 scalar; `1.25 * np.float32` (python float x NumPy scalar) is float64 there, so product and sum are float64.  (Written with
    explicit casts because NumPy >= 2, installed here, would keep the expression in float32.)"""
    s = np.asarray(maxvals, np.float32)
    return float(np.float64(np.mean(s)) + 1.25 * np.float64(np.max(s)))


# --------------------------------------------------------------------------
# a8' soft-arg-max decode (L1JointRegression configs only)
# --------------------------------------------------------------------------

def softargmax_decode(hm: np.ndarray, bbox, norm_type: str = "softmax"):
    """heatmap_to_coord_simple_regress (transforms.py:586-642) for one item.

    Computed in float32 like the reference's torch path, affine in float64.
    """
    J, H, W = hm.shape
    x = hm.reshape(J, -1).astype(np.float32)
    if norm_type == "softmax":
        e = np.exp(x - x.max(axis=1, keepdims=True))
        p = e / e.sum(axis=1, keepdims=True)
        score = np.ones((J, 1), np.float32)
    elif norm_type == "sigmoid":
        p = (1.0 / (1.0 + np.exp(-x))).astype(np.float32)
        score = p.max(axis=1, keepdims=True)
    elif norm_type == "divide_sum":
        p = x / x.sum(axis=1, keepdims=True)
        score = np.ones((J, 1), np.float32)
    else:
        raise NotImplementedError(norm_type)
    p = (p / p.sum(axis=1, keepdims=True)).reshape(J, H, W)
    ex = (p.sum(axis=1) * np.arange(W, dtype=np.float32)).sum(axis=1)
    ey = (p.sum(axis=2) * np.arange(H, dtype=np.float32)).sum(axis=1)
    u = ((ex / np.float32(W) - np.float32(0.5)) + np.float32(0.5)) * np.float32(W)
    v = ((ey / np.float32(H) - np.float32(0.5)) + np.float32(0.5)) * np.float32(H)
    T = crop_to_image_affine(bbox, W, H)
    out = np.zeros((J, 2), np.float32)
    for j in range(J):
        out[j] = T @ np.array([float(u[j]), float(v[j]), 1.0])
    return dict(coords=out, maxvals=score.astype(np.float32))


# --------------------------------------------------------------------------
# a9 / a10  temporal heat-map / pose continuity      ActiveLearning.py
# --------------------------------------------------------------------------

def thc_pair(a: np.ndarray, b: np.ndarray, norm_type: str = "L1") -> np.float32:
    """compute_thc (ActiveLearning.py:747-760): sum|a-b| / J or sum (a-b)^2 / J."""
    J = a.shape[0]
    if norm_type == "L1":
        return np.sum(np.abs(a - b)) / J
    if norm_type == "L2":
        return np.sum(np.square(a - b)) / J
    raise ValueError(norm_type)


def combine_neighbours(v_prev, v_next, is_prev: bool, is_next: bool):
    """The isPrev/isNext rule shared by THC and TPC (ActiveLearning.py:333-363):
    sum over existing neighbours, doubled when exactly one exists, 0 when none."""
    t = 0
    if is_prev:
        t = t + v_prev
    if is_next:
        t = t + v_next
        if not is_prev:
            t = t * 2
    elif is_prev:
        t = t * 2
    return t


def thc_item(cur, prev, nxt, is_prev, is_next, norm_type="L1") -> float:
    vp = thc_pair(cur, prev, norm_type) if is_prev else 0
    vn = thc_pair(cur, nxt, norm_type) if is_next else 0
    return float(combine_neighbours(vp, vn, bool(is_prev), bool(is_next)))


def tpc_pair(cur_coords: np.ndarray, adj_hm: np.ndarray, bbox, thresh: float) -> int:
    """compute_tpc (ActiveLearning.py:736-745): decode the neighbour with the
    CURRENT box, count joints displaced by more than ``thresh``."""
    adj = decode_heatmaps(adj_hm, bbox)["coords"]
    dist = np.linalg.norm(cur_coords - adj, axis=1)
    return int(np.count_nonzero(dist > thresh))


def tpc_item(cur_hm, prev_hm, next_hm, bbox, is_prev, is_next) -> float:
    """TPC branch of eval_and_query (ActiveLearning.py:333-344)."""
    thresh = 0.01 * np.sqrt((bbox[2] - bbox[0]) * (bbox[3] - bbox[1]))
    cur = decode_heatmaps(cur_hm, bbox)["coords"]
    vp = tpc_pair(cur, prev_hm, bbox, thresh) if is_prev else 0
    vn = tpc_pair(cur, next_hm, bbox, thresh) if is_next else 0
    return float(combine_neighbours(vp, vn, bool(is_prev), bool(is_next)))


# --------------------------------------------------------------------------
# a11 local-peak mean                               active_learning/local_peak.py
# --------------------------------------------------------------------------

def _maxfilter3x3_zero(img: np.ndarray) -> np.ndarray:
    """3x3 maximum filter with constant-0 border (scipy maximum_filter,
    footprint ones(3,3), mode='constant', cval=0; local_peak.py:6)."""
    H, W = img.shape
    pad = np.zeros((H + 2, W + 2), img.dtype)
    pad[1:-1, 1:-1] = img
    out = pad[1:-1, 1:-1].copy()
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            np.maximum(out, pad[dy:dy + H, dx:dx + W], out=out)
    return out


def localpeak_values(img: np.ndarray, order: float = 0.5) -> np.ndarray:
    """local_peak.py:5-10.  Peaks = pixels equal to their zero-padded 3x3 max;
    kept = peaks >= order * (largest peak).  Row-major order."""
    is_peak = img == _maxfilter3x3_zero(img)
    if not is_peak.any():
        return img[is_peak]                        # empty
    top = img[is_peak].max()
    keep = is_peak & (img >= top * order)
    return img[keep]


def localpeak_stats(hm: np.ndarray, order: float = 0.5):
    """Per-joint kept-peak count (int64 (J,)) and float64 sum, plus the mean."""
    cnt = np.zeros(hm.shape[0], np.int64)
    vals = []
    for j, img in enumerate(hm):
        v = localpeak_values(img, order)
        cnt[j] = v.size
        vals.append(v)
    allv = np.hstack(vals)
    with np.errstate(invalid="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            mean = allv.mean() if allv.size else np.float32(np.nan)
    return cnt, allv, mean


def localpeak_mean(hm: np.ndarray, order: float = 0.5):
    """local_peak.py:12-22: mean over the concatenated kept peaks of all joints
    (nan when there is none)."""
    return localpeak_stats(hm, order)[2]


# --------------------------------------------------------------------------
# a12 whole-body pose unnaturalness     active_learning/Whole_body_AE/*
# --------------------------------------------------------------------------

_TRIANGLES = np.array([[8, 6, 12], [6, 8, 10], [5, 7, 9], [7, 5, 11],
                       [11, 12, 14], [12, 11, 13], [12, 14, 16], [11, 13, 15]])


def xyxy_to_xywh(b):
    """alphapose/utils/bbox.py:91-97 (inclusive-pixel convention: +1)."""
    return (b[0], b[1], b[2] - b[0] + 1, b[3] - b[1] + 1)


def hybrid_feature(bbox_xywh, kpts51) -> np.ndarray:
    """compute_hybrid (hybrid_feature.py:14-59): 17 + 17 + 8 = 42 float64."""
    k = np.asarray(kpts51, np.float64)
    x, y, s = k[0::3], k[1::3], k[2::3]
    height = float(bbox_xywh[3])
    assert height > 0, "height of human body must be positive!"
    assert s.sum() > 0, "at least one visible keypoint is required!"
    gx = (x * s).sum() / s.sum()
    gy = (y * s).sum() / s.sum()
    eps = 1e-6
    a, b, c = _TRIANGLES[:, 0], _TRIANGLES[:, 1], _TRIANGLES[:, 2]
    m1 = (y[b] - y[a]) / (x[b] - x[a] + eps)
    m2 = (y[c] - y[b]) / (x[c] - x[b] + eps)
    ang = np.arctan(np.abs((m1 - m2) / (1 + m1 * m2 + eps)))
    return np.hstack(((x - gx) / height, (y - gy) / height, ang))


def ae_forward(feat32: np.ndarray, weights: dict) -> np.ndarray:
    """WholeBodyAE.forward (AutoEncoder.py:13-39) in float32 numpy.

    ``weights``: state-dict style ``encoder.{0,2,4,6}.{weight,bias}``,
    ``decoder.{0,2,4,6}.{weight,bias}``.  ReLU between layers, none after the
    bottleneck Linear, Sigmoid after the last.
    """
    h = feat32.astype(np.float32)
    for i in (0, 2, 4, 6):
        h = weights[f"encoder.{i}.weight"].astype(np.float32) @ h + weights[f"encoder.{i}.bias"].astype(np.float32)
        if i != 6:
            h = np.maximum(h, 0)
    for i in (0, 2, 4, 6):
        h = weights[f"decoder.{i}.weight"].astype(np.float32) @ h + weights[f"decoder.{i}.bias"].astype(np.float32)
        if i != 6:
            h = np.maximum(h, 0)
    return (1.0 / (1.0 + np.exp(-h.astype(np.float32)))).astype(np.float32)


_WPU38 = np.r_[0:3, 5:20, 22:42]


def wpu_item(bbox_crop_xyxy, kpts51, weights, only38: bool = False) -> float:
    """WPU of one item.  THC+WPU branch (ActiveLearning.py:364-370) uses all 42
    values; the WPU-only branch (:371-386) drops indices 3,4,20,21 from both
    the input and the reconstruction before the MSE."""
    f = hybrid_feature(xyxy_to_xywh(bbox_crop_xyxy), kpts51).astype(np.float32)
    r = ae_forward(f, weights)
    if only38:
        f, r = f[_WPU38], r[_WPU38]
    return float(np.mean((r - f) ** 2, dtype=np.float32))


# --------------------------------------------------------------------------
# a6 masked MSE, a7 AdamW                       ActiveLearning.py:669, :224-231
# --------------------------------------------------------------------------

def masked_mse(out: np.ndarray, tgt: np.ndarray, mask: np.ndarray):
    """loss = 0.5 * mean((out*m - tgt*m)^2), grad wrt out (ActiveLearning.py:669).
    ``mask`` broadcasts as (B,J,1,1).  float64 accumulation for the scalar."""
    m = mask.reshape(mask.shape[0], mask.shape[1], 1, 1).astype(np.float32)
    d = out * m - tgt * m
    loss = 0.5 * float(np.mean(d.astype(np.float64) ** 2))
    grad = (d * m / np.float32(d.size)).astype(np.float32)
    return loss, grad


def adamw_step(p, g, m, v, step: int, lr: float, wd: float,
               beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor update (decoupled decay), float32 state.
    Returns new (p, m, v).  ``step`` is the 1-based step count."""
    f = np.float32
    p = p * f(1.0 - lr * wd)
    m = m + (g - m) * f(1.0 - beta1)
    v = v * f(beta2) + g * g * f(1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / f(np.sqrt(bc2)) + f(eps)
    p = p - f(lr / bc1) * (m / denom)
    return p.astype(f), m.astype(f), v.astype(f)


def adam_step(p, g, m, v, step: int, lr: float, wd: float = 0.0, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update (ActiveLearning.py:222-223): L2 decay enters the gradient."""
    f = np.float32
    g = g + f(wd) * p
    m = m + (g - m) * f(1.0 - beta1)
    v = v * f(beta2) + g * g * f(1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    p = p - f(lr / bc1) * (m / (np.sqrt(v) / f(np.sqrt(bc2)) + f(eps)))
    return p.astype(f), m.astype(f), v.astype(f)


def sgd_step(p, g, buf, step: int, lr: float, momentum: float = 0.9, wd: float = 0.0):
    """torch.optim.SGD with momentum, dampening 0, no Nesterov (ActiveLearning.py:220-221)."""
    f = np.float32
    g = g + f(wd) * p
    buf = g.copy() if step == 1 else buf * f(momentum) + g
    p = p - f(lr) * buf
    return p.astype(f), buf.astype(f)


# --------------------------------------------------------------------------
# §8f rank 2 (part): Gaussian heat-map targets        simple_transform.py:122-158
# --------------------------------------------------------------------------

def target_generator(joints_xy: np.ndarray, vis: np.ndarray, hm_hw=(64, 48), in_hw=(256, 192), sigma: float = 2.0):
    """SimpleTransform._target_generator for one person: joints_xy (J,2) float32 input-pixel coordinates, vis (J,)
    -> target (J,H,W) float32, target_weight (J,1,1) float32."""
    H, W = hm_hw
    J = joints_xy.shape[0]
    # _feat_stride = input_size / heatmap_size with both (H, W); the reference applies index 0 to x and index 1 to y
    # (simple_transform.py:130-131) — the two are equal (4.0) for every preset
    stride = np.array(in_hw, np.float64) / np.array(hm_hw, np.float64)
    weight = np.ones((J, 1), np.float32)
    weight[:, 0] = vis
    target = np.zeros((J, H, W), np.float32)
    tmp = sigma * 3
    for i in range(J):
        mu_x = int(joints_xy[i, 0] / stride[0] + 0.5)
        mu_y = int(joints_xy[i, 1] / stride[1] + 0.5)
        ul = [int(mu_x - tmp), int(mu_y - tmp)]
        br = [int(mu_x + tmp + 1), int(mu_y + tmp + 1)]
        if ul[0] >= W or ul[1] >= H or br[0] < 0 or br[1] < 0:
            weight[i] = 0
            continue
        size = 2 * tmp + 1
        x = np.arange(0, size, 1, np.float32)
        y = x[:, np.newaxis]
        x0 = y0 = size // 2
        g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * (sigma ** 2)))
        g_x = max(0, -ul[0]), min(br[0], W) - ul[0]
        g_y = max(0, -ul[1]), min(br[1], H) - ul[1]
        img_x = max(0, ul[0]), min(br[0], W)
        img_y = max(0, ul[1]), min(br[1], H)
        if weight[i] > 0.5:
            target[i, img_y[0]:img_y[1], img_x[0]:img_x[1]] = g[g_y[0]:g_y[1], g_x[0]:g_x[1]]
    return target, np.expand_dims(weight, -1)


# --------------------------------------------------------------------------
# a13: MPE / Margin / Entropy          ActiveLearning.py:387-396, 762-796
# --------------------------------------------------------------------------
# `peak_local_max` is scikit-image's (the reference pins scikit-image==0.24.0, requirements.txt:172).  The package
# does not import in this image, but the SOURCE of scikit-image 0.18.3 sits at /opt/conda/lib/python3.9/site-packages/
# skimage and `feature/peak.py` + `_shared/coord.py` are pure Python: `tools/make_golden.py --only peaks` executes them
# from there and freezes their answers (and the reference's compute_mpe / compute_margin run on top of them) in
# tests/golden/peaks.npz.  PINNED on that fixture: positions, order and counts are identical on every plane whose
# candidate maxima have pairwise distinct values (392 of 408 network-like planes, all border / spacing / truncation /
# degenerate cases, the inputs of scikit-image's own test_peak.py) — tests/test_oracle_golden.py.
# VERSION-DEPENDENT, and therefore not part of the parity claim: the order among candidates of EXACTLY equal value.
# 0.18.3 sorts them with `np.argsort(-intensities)` (introsort, unstable beyond 16 elements, peak.py:17); 0.24 uses
# kind="stable", i.e. row-major order among equals, which is what the function below restates (maximum filter over a
# (2*min_distance+1)^2 footprint, threshold = image.min(), border of width min_distance excluded, candidates by
# descending intensity, `ensure_spacing` = greedy rejection of candidates at Chebyshev distance < min_distance from an
# accepted one, first `num_peaks` kept).  On the 16 tied planes of the fixture 12 agree with 0.18.3 and 4 pick another
# of the equal maxima; MPE / Margin are unchanged by that choice there (equal values).  The filter's border mode
# ('constant' in 0.18.3, 'nearest' in 0.24) cannot matter: every pixel whose window leaves the image is in the excluded
# border.  softmax / entropy are scipy's (present: pinned directly).

def peak_local_max_5(img: np.ndarray, min_distance: int = 5, num_peaks: int = 5) -> np.ndarray:
    """(H,W) -> (k,2) int rows of (row, col), k <= num_peaks, highest peaks first."""
    from scipy import ndimage
    size = 2 * min_distance + 1
    mx = ndimage.maximum_filter(img, footprint=np.ones((size, size), bool), mode="nearest")
    out = img == mx
    if np.all(out):
        out[:] = False
    out &= img > img.min()
    b = min_distance
    out[:b, :] = False; out[-b:, :] = False; out[:, :b] = False; out[:, -b:] = False
    rows, cols = np.nonzero(out)
    order = np.argsort(-img[rows, cols], kind="stable")
    coords = np.stack([rows[order], cols[order]], 1)
    kept = []
    for c in coords:
        if all(max(abs(int(c[0]) - k[0]), abs(int(c[1]) - k[1])) >= min_distance for k in kept):
            kept.append((int(c[0]), int(c[1])))
            if len(kept) == num_peaks:
                break
    return np.asarray(kept, np.int64).reshape(-1, 2)


def mpe_item(hm: np.ndarray) -> float:
    """compute_mpe (ActiveLearning.py:762-778): sum over joints of entropy(softmax(values of <= 5 local peaks))."""
    from scipy.special import softmax
    from scipy.stats import entropy
    total = 0
    for h in hm:
        loc = peak_local_max_5(h)
        peaks = h[loc[:, 0], loc[:, 1]]
        if peaks.shape[0] > 0:
            total += entropy(softmax(peaks))
    return float(total)


def margin_item(hm: np.ndarray) -> float:
    """compute_margin (ActiveLearning.py:780-788): sum over joints of |top peak - second peak|."""
    total = 0
    for h in hm:
        loc = peak_local_max_5(h)
        peaks = h[loc[:, 0], loc[:, 1]]
        if peaks.shape[0] > 1:
            total += np.linalg.norm(peaks[0] - peaks[1])
    return float(total)


def entropy_item(hm: np.ndarray) -> float:
    """compute_entropy (ActiveLearning.py:790-796): sum over joints of scipy.stats.entropy(flattened map).  scipy
    normalises by the sum and uses entr(): -inf for a negative entry, nan for a zero sum — kept as is."""
    from scipy.stats import entropy
    total = 0
    with np.errstate(all="ignore"):
        for h in hm:
            total += entropy(h.flatten())
    return float(total)


# --------------------------------------------------------------------------
# §8f rank 1: OKS and heat-map accuracy         al_metric.py:42-69, metrics.py:118-245
# --------------------------------------------------------------------------

_OKS_SIG = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62,
                     1.07, 1.07, .87, .87, .89, .89]) / 10.0
_OKS_VAR = (_OKS_SIG * 2) ** 2


def oks(bbox_xywh, pred51, gt51) -> float:
    """compute_OKS (al_metric.py:42-69) with the box area as the scale."""
    d = np.asarray(pred51, np.float64)
    g = np.asarray(gt51, np.float64)
    xg, yg, vg = g[0::3], g[1::3], g[2::3]
    xd, yd = d[0::3], d[1::3]
    bx, by, bw, bh = [float(t) for t in bbox_xywh]
    vis = vg > 0
    if vis.any():
        dx, dy = xd - xg, yd - yg
    else:
        z = np.zeros(17)
        dx = np.maximum(z, (bx - bw) - xd) + np.maximum(z, xd - (bx + 2 * bw))
        dy = np.maximum(z, (by - bh) - yd) + np.maximum(z, yd - (by + 2 * bh))
    e = (dx ** 2 + dy ** 2) / _OKS_VAR / (bw * bh + np.spacing(1)) * 0.5
    if vis.any():
        e = e[vis]
    return float(np.sum(np.exp(-e)) / e.shape[0])


def heatmap_accuracy(pred: np.ndarray, label: np.ndarray, thr: float = 0.5) -> float:
    """calc_accuracy (metrics.py:118-147) for (B,J,H,W) float32 arrays."""
    B, J, H, W = pred.shape

    def peaks(a):
        flat = a.reshape(B, J, -1)
        idx = flat.argmax(axis=2)
        mv = np.take_along_axis(flat, idx[..., None], axis=2)[..., 0]
        xy = np.stack([(idx % W), (idx // W)], axis=2).astype(np.float32)
        return xy * (mv > 0)[..., None].astype(np.float32)

    p, t = peaks(pred), peaks(label)
    norm = np.array([W, H], np.float64) / 10
    dist = np.zeros((J, B))
    for n in range(B):
        for c in range(J):
            if t[n, c, 0] > 1 and t[n, c, 1] > 1:
                dist[c, n] = np.linalg.norm(p[n, c] / norm - t[n, c] / norm)
    total, cnt = 0.0, 0
    for c in range(J):
        used = dist[c] != 0
        if used.sum() > 0:
            total += float((dist[c][used] < thr).sum()) / used.sum()
            cnt += 1
    return total / cnt if cnt else 0
