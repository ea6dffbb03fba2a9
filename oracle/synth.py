"""Seeded, platform-stable synthetic weights and inputs (TEST ORACLE support).

Everything is drawn from ``numpy.random.RandomState`` (legacy generator, frozen
bit-stream) so that the golden fixtures, the CPU tests and the GPU-box tests
regenerate identical tensors from a seed instead of shipping 136 MB state
dicts.  Shapes follow SURVEY.md §8(d): crops ``rand - (0.406, 0.457, 0.480)``
(simple_transform.py:246-249), boxes ``[100, 50, 100+w, 50+4w/3]``.
"""
from __future__ import annotations

import zlib

import numpy as np

SEED = 166          # the reference's own seed (scripts/Run_active_learning.py:113)
PIXEL_MEAN = (0.406, 0.457, 0.480)


def _rs(seed: int, key: str) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def tensor_for(key: str, shape, seed: int = SEED) -> np.ndarray:
    """One state-dict entry, chosen by key suffix / rank (float32, or int64 for
    ``num_batches_tracked``)."""
    r = _rs(seed, key)
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return np.zeros(shape, np.int64)
    if key.endswith("running_var"):
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if key.endswith("running_mean"):
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 1 and (key.endswith("bn3.weight") or (".branches." in key and key.endswith("bn2.weight"))
                            or (".fuse_layers." in key and key.endswith(".weight"))):
        # last BN of a residual / fusion branch: keep the un-normalised (eval-mode) network from blowing up
        return r.uniform(0.2, 0.4, shape).astype(np.float32)
    if len(shape) == 1 and key.endswith("weight"):            # BN gamma
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if len(shape) == 1:                                        # any bias / BN beta
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 4:                                        # conv OIHW or deconv IOHW
        if "deconv" in key:
            fan_in = shape[0] * shape[2] * shape[3] / 4.0      # 2x2 taps reach each output
        else:
            fan_in = shape[1] * shape[2] * shape[3]
        return (np.sqrt(2.0 / fan_in) * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 2:                                        # Linear (out, in)
        return (np.sqrt(1.0 / shape[1]) * r.standard_normal(shape)).astype(np.float32)
    return r.standard_normal(shape).astype(np.float32)


def state_dict_for(module, seed: int = SEED) -> dict:
    """Synthetic state dict for any torch module (keys/shapes from the module)."""
    import torch
    return {k: torch.from_numpy(tensor_for(k, v.shape, seed)) for k, v in module.state_dict().items()}


def crops(n: int, seed: int = SEED, hw=(256, 192)) -> np.ndarray:
    r = _rs(seed, f"crops{n}x{hw}")
    x = r.random_sample((n, 3, hw[0], hw[1])).astype(np.float32)
    x -= np.asarray(PIXEL_MEAN, np.float32).reshape(1, 3, 1, 1)
    return x


def bboxes(n: int, seed: int = SEED) -> np.ndarray:
    r = _rs(seed, f"bboxes{n}")
    w = r.uniform(60, 240, n)
    return np.stack([np.full(n, 100.0), np.full(n, 50.0), 100.0 + w, 50.0 + w * 4.0 / 3.0], 1).astype(np.float32)


def video_flags(n: int, tracks: int):
    """isPrev / isNext for ``tracks`` equal-length tracks laid out contiguously
    in id order (posetrack21.py:148-178 semantics: neighbour has same track)."""
    per = n // tracks
    pos = np.arange(n) % per
    return (pos != 0), (pos != per - 1)


def blob_heatmaps(n: int, seed: int = SEED, J: int = 17, hw=(64, 48), noise: float = 0.02) -> np.ndarray:
    """Heat-maps that look like a pose network's: 1-3 Gaussian bumps per joint
    (sigma 2, main amplitude 0.3..1) over low-amplitude noise; consecutive
    items drift slowly so THC/TPC see realistic neighbour differences."""
    r = _rs(seed, f"blobs{n}x{J}x{hw}")
    H, W = hw
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    out = np.empty((n, J, H, W), np.float32)
    cx = r.uniform(4, W - 4, J)
    cy = r.uniform(4, H - 4, J)
    for i in range(n):
        cx = np.clip(cx + r.normal(0, 0.7, J), 1, W - 2)
        cy = np.clip(cy + r.normal(0, 0.7, J), 1, H - 2)
        for j in range(J):
            amp = r.uniform(0.3, 1.0)
            m = amp * np.exp(-((xx - cx[j]) ** 2 + (yy - cy[j]) ** 2) / 8.0)
            for _ in range(r.randint(0, 3)):
                m += r.uniform(0.1, 0.8) * amp * np.exp(
                    -((xx - r.uniform(0, W)) ** 2 + (yy - r.uniform(0, H)) ** 2) / 8.0)
            out[i, j] = m + noise * r.standard_normal((H, W))
    return out


def peak_items(n: int = 24, seed: int = 77) -> np.ndarray:
    """Inputs of tests/golden/peaks.npz (MPE / Margin against the real scikit-image): network-like maps, one
    non-negative item, one with a flat zero background, one all-negative, one value-quantised (ties among candidates)."""
    hm = blob_heatmaps(n, seed=seed)
    hm[3] = np.abs(hm[3]) + 1e-3
    hm[7] = blob_heatmaps(1, seed=seed + 1, noise=0.0)[0]
    hm[11] = -np.abs(hm[11]) - 0.1
    hm[13] = np.round(hm[13] * 4) / 4
    return hm


def gaussian_targets(n: int, seed: int = SEED, J: int = 17, hw=(64, 48), sigma: float = 2.0, p_zero: float = 0.2):
    """Labels like SimpleTransform._target_generator (simple_transform.py:
    122-158): a (6*sigma+3)^2 Gaussian patch at an integer joint position;
    ``mask`` (n,J,1,1) with about ``p_zero`` zeros."""
    r = _rs(seed, f"targets{n}x{J}x{hw}")
    H, W = hw
    t = np.zeros((n, J, H, W), np.float32)
    rad = int(3 * sigma)
    g = np.arange(-rad, rad + 1, dtype=np.float32)
    patch = np.exp(-(g[None, :] ** 2 + g[:, None] ** 2) / (2 * sigma ** 2))
    for i in range(n):
        for j in range(J):
            mx, my = r.randint(0, W), r.randint(0, H)
            x0, x1 = max(0, mx - rad), min(W, mx + rad + 1)
            y0, y1 = max(0, my - rad), min(H, my + rad + 1)
            t[i, j, y0:y1, x0:x1] = patch[y0 - (my - rad):y1 - (my - rad), x0 - (mx - rad):x1 - (mx - rad)]
    mask = (r.random_sample((n, J, 1, 1)) >= p_zero).astype(np.float32)
    return t, mask


def l1_inputs(norm: str):
    """Seeded inputs of the L1JointRegression fixtures (tools/make_golden.py gen_l1_loss and the tests)."""
    hm = blob_heatmaps(3, seed=31) * 12.0                            # peaked after the soft-max
    if norm == "divide_sum":
        hm = np.abs(hm) + 0.01
    r = np.random.RandomState(77)
    gt = r.uniform(-0.5, 0.5, (3, 34)).astype(np.float32)
    vis = np.repeat((r.random_sample((3, 17)) > 0.25).astype(np.float32), 2, axis=1)
    return hm.astype(np.float32), gt, vis


def target_joints(n: int, hm_hw, in_hw, seed: int = 41, J: int = 17):
    """Seeded joint positions (input-pixel coordinates) and visibility flags for the target-generator fixtures: most
    inside the crop, some on / beyond the borders (patch clipped or missed entirely), ~20 % invisible."""
    r = _rs(seed, f"tjoints{n}x{hm_hw}")
    xy = np.stack([r.uniform(-40, in_hw[1] + 40, (n, J)), r.uniform(-40, in_hw[0] + 40, (n, J))], 2).astype(np.float32)
    xy[0, 0] = [0.0, 0.0]; xy[0, 1] = [in_hw[1] - 1, in_hw[0] - 1]; xy[0, 2] = [-30.0, 10.0]; xy[0, 3] = [in_hw[1] + 27.9, 5.0]
    vis = (r.random_sample((n, J)) >= 0.2).astype(np.float32)
    return xy, vis


def crop_cases(n: int, seed: int = 57):
    """Seeded person boxes [xmin,ymin,xmax,ymax] (float64 python-float semantics, like annotation json) inside a 640x480
    frame — wide, tall, tiny, partly outside the frame — plus a rotation per case (half of them 0, as at test time)."""
    r = _rs(seed, f"cropcases{n}")
    x0 = r.uniform(-40, 520, n); y0 = r.uniform(-30, 380, n)
    w = r.uniform(8, 300, n); h = r.uniform(8, 420, n)
    box = np.stack([x0, y0, x0 + w, y0 + h], 1)
    box[0] = [100.0, 50.0, 292.0, 306.0]                     # exactly the 3:4 aspect
    box[1] = [10.5, 20.25, 600.0, 60.0]                      # very wide
    rot = np.where(r.random_sample(n) < 0.5, 0.0, np.clip(r.randn(n) * 40, -80, 80))
    rot[:2] = 0.0
    return box, rot


def u8_frame(h: int = 480, w: int = 640, seed: int = 58) -> np.ndarray:
    """Seeded (h,w,3) uint8 video frame: smooth colour gradients + a checker pattern + noise (exercises every sub-pixel
    phase of the bilinear taps and the constant border)."""
    r = _rs(seed, f"frame{h}x{w}")
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 / max(w - 1, 1)), (yy * 255 / max(h - 1, 1)), ((xx // 16 + yy // 16) % 2) * 200 + 20], 2)
    img = base + r.randint(-20, 21, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def frame_video(n_frames: int = 8, tracks: int = 2, hw=(240, 320), seed: int = 71, J: int = 17):
    """Seeded stand-in for a decoded PoseTrack video: ``n_frames`` uint8 RGB frames and one annotation per (track, frame)
    in the field layout the reference's loaders build (posetrack21.py:103-115): a person box drifting through the
    frame, joints inside it, ~15 % invisible; ``id`` orders items track by track, frame by frame."""
    r = _rs(seed, f"video{n_frames}x{tracks}")
    frames = [u8_frame(hw[0], hw[1], seed + 1 + f) for f in range(n_frames)]
    anns = []
    for t in range(tracks):
        x0, y0 = r.uniform(10, hw[1] * 0.4), r.uniform(5, hw[0] * 0.2)
        w, h = r.uniform(50, 110), r.uniform(110, 170)
        for f in range(n_frames):
            bx, by = x0 + 3.5 * f + r.uniform(-1, 1), y0 + 1.25 * f + r.uniform(-1, 1)
            j3 = np.zeros((J, 3, 2), np.float32)
            j3[:, 0, 0] = r.uniform(bx, bx + w, J)
            j3[:, 1, 0] = r.uniform(by, by + h, J)
            vis = (r.random_sample(J) > 0.15).astype(np.float32)
            j3[:, 0, 1] = j3[:, 1, 1] = vis
            j3[:, :, 0] *= j3[:, :, 1]
            kp = np.stack([j3[:, 0, 0], j3[:, 1, 0], vis], 1).reshape(-1).astype(np.float32)
            anns.append({"bbox": (float(bx), float(by), float(bx + w), float(by + h)), "joints_3d": j3, "keypoint": kp.tolist(),
                         "id": t * 1000 + f, "ann_id": 500000 + t * 1000 + f, "img_id": 9000 + f, "track_id": f"v{t}", "frame": f})
    return frames, anns


def write_coco_video(root: str, n_frames: int = 4, tracks: int = 2, hw=(120, 160), seed: int = 73, fmt: str = "posetrack"):
    """Write a tiny PoseTrack21- / JRDB-style dataset under ``root``: lossless PNG frames and one COCO-format json with the
    fields the reference's loaders read (images: id, image_id, vid_id, file_name, width, height; annotations: id, image_id,
    track_id, bbox xywh, keypoints).  Includes a person with a zero-area box, one without key-points and one with no
    visible joint (all three must be dropped).  -> (annotation path relative to root, frames, expected kept annotation ids)."""
    import json
    import os
    from PIL import Image
    r = _rs(seed, f"coco{n_frames}x{tracks}{fmt}")
    os.makedirs(os.path.join(root, "images", "vid0"), exist_ok=True)
    frames, images, anns, kept = [], [], [], []
    mul = 100 if fmt == "posetrack" else 1000                        # annotation id = image id * 100 (1000) + person index, as in the real files
    for f in range(n_frames):
        img = u8_frame(hw[0], hw[1], seed + f)
        name = os.path.join("images", "vid0", f"{f:06d}.png")
        Image.fromarray(img).save(os.path.join(root, name))
        frames.append(img)
        image_id = 1000200 + f
        images.append({"id": image_id, "image_id": image_id, "vid_id": 7, "file_name": name, "width": hw[1], "height": hw[0]})
        for t in range(tracks):
            x, y, w, h = 10.0 + 30 * t + 2 * f, 8.0 + f, 40.0 + t, 70.0
            kp = []
            for j in range(17):
                kp += [float(r.uniform(x, x + w)), float(r.uniform(y, y + h)), int(r.random_sample() > 0.2)]
            kp[2] = 1
            anns.append({"id": image_id * mul + t, "image_id": image_id, "track_id": t, "bbox": [x, y, w, h], "keypoints": kp, "category_id": 1})
            kept.append(image_id * mul + t)
        if f == 1:                                                  # three invalid persons
            anns.append({"id": image_id * mul + 50, "image_id": image_id, "track_id": 50, "bbox": [5.0, 5.0, 1.0, 1.0], "keypoints": [1.0, 1.0, 1] * 17, "category_id": 1})
            anns.append({"id": image_id * mul + 51, "image_id": image_id, "track_id": 51, "bbox": [5.0, 5.0, 20.0, 20.0], "keypoints": [0, 0, 0] * 17, "category_id": 1})
            anns.append({"id": image_id * mul + 52, "image_id": image_id, "track_id": 52, "bbox": [5.0, 5.0, 20.0, 20.0], "keypoints": [3.0, 4.0, 0] * 17, "category_id": 1})
    ann_path = os.path.join("annotations", "val.json")
    os.makedirs(os.path.join(root, "annotations"), exist_ok=True)
    with open(os.path.join(root, ann_path), "w") as fh:
        json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "person"}]}, fh)
    return ann_path, frames, kept
