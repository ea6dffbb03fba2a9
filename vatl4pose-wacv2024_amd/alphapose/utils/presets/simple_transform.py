"""Person-crop producer with the reference's ``SimpleTransform`` interface, warping on MI355X.

Reference: alphapose/utils/presets/simple_transform.py — ``__init__`` :54-79, ``test_transform`` :81-98,
``_target_generator`` :122-158, ``_integral_target_generator`` :160-177, ``__call__`` :179-251,
``half_body_transform`` :253-304.  The per-item methods keep their signatures and results; the pixel work
(cv2.warpAffine + im_to_torch + mean shift, Gaussian targets) is one ``libvatl_hip.so`` launch each, and
``crop_batch`` / ``targets_batch`` do a whole batch per launch.  The geometry
(centre/scale, augmentation draws, the 2x3 matrix) stays on the host in numpy like the reference's: it is a few
float64 operations per crop and its random draws (``np.random`` / ``random``) must consume the same streams.
"""
from __future__ import annotations

import random

import numpy as np
import torch

import vatl_hip as vh

from ..bbox import (_box_to_center_scale, _center_scale_to_box, box_to_center_scale_batch, center_scale_to_box_batch)
from ..transforms import flip_joints_3d, get_affine_transform_batch, invert_affine_batch


def _device():
    if not torch.cuda.is_available():
        raise vh.VatlError("SimpleTransform crops on MI355X only (there is deliberately no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


class FrameArena:
    """uint8 frames packed back to back in one device buffer — what ``vatl_crop_warp_affine`` reads."""

    def __init__(self, frames, device=None):
        dev = device or _device()
        frames = [np.ascontiguousarray(f) for f in frames]
        for f in frames:
            if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
                raise ValueError(f"frames must be (h, w, 3) uint8, got {f.dtype} {f.shape}")
        sizes = np.array([f.size for f in frames], np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        self.hw = np.array([f.shape[:2] for f in frames], np.int32).reshape(-1, 2)
        host = torch.from_numpy(np.concatenate([f.reshape(-1) for f in frames])) if frames else torch.empty(0, dtype=torch.uint8)
        self.data = host.to(dev, non_blocking=False)


class SimpleTransform(object):
    """Generation of cropped input person and pose heat-maps (constructor arguments as the reference's, :54)."""

    def __init__(self, dataset, scale_factor, add_dpg, input_size, output_size, rot, sigma, train, gpu_device=None, loss_type="MSELoss"):
        inp_h, inp_w = input_size
        self._input_size, self._heatmap_size = input_size, output_size
        self._aspect_ratio = float(inp_w) / inp_h                       # crops are w / h = 3 / 4
        self._feat_stride = np.asarray(input_size) / np.asarray(output_size)
        self._train, self._loss_type, self._sigma = train, loss_type, sigma
        self._scale_factor, self._rot, self._add_dpg, self._gpu_device = scale_factor, rot, add_dpg, gpu_device
        self._joint_pairs = dataset.joint_pairs
        self.pixel_std = 1
        if train:                                                       # what the half-body augmentation reads from the dataset (:70-75)
            for name in ("num_joints_half_body", "prob_half_body", "upper_body_ids", "lower_body_ids"):
                setattr(self, name, getattr(dataset, name))

    # ------------------------------------------------------------------ batched fast path
    def crop_batch(self, arena: FrameArena, frame_index, centers, scales, rots=0.0, mirror=None, out=None):
        """Warp B crops in one launch: crop b reads frame ``frame_index[b]`` of ``arena`` through the matrix
        get_affine_transform(centers[b], scales[b], rots[b], [inp_w, inp_h]) -> (B,3,inp_h,inp_w) fp32 device tensor."""
        inp_h, inp_w = int(self._input_size[0]), int(self._input_size[1])
        fi = np.asarray(frame_index, np.int64).reshape(-1)
        n = fi.shape[0]
        trans = get_affine_transform_batch(centers, scales, rots, [inp_w, inp_h])
        minv = invert_affine_batch(trans)
        hwf = np.zeros((n, 3), np.int32)
        hwf[:, :2] = arena.hw[fi]
        if mirror is not None:
            hwf[:, 2] = np.asarray(mirror).astype(np.int32)
        dev = arena.data.device
        crops, _ = vh.crop_warp_affine(arena.data, vh.upload(arena.offsets[fi], dev), vh.upload(hwf, dev),
                                       vh.upload(np.ascontiguousarray(minv), dev), (inp_h, inp_w), out=out)
        return crops, trans

    def test_transform_batch(self, arena: FrameArena, frame_index, boxes_xyxy):
        """``test_transform`` (:81-98) for B (frame, box) pairs -> crops (B,3,H,W) on device, boxes (B,4) float32 tensor."""
        centers, scales = box_to_center_scale_batch(boxes_xyxy, self._aspect_ratio)
        crops, _ = self.crop_batch(arena, frame_index, centers, scales, 0.0)
        return crops, torch.from_numpy(center_scale_to_box_batch(centers, scales).astype(np.float32))

    def targets_batch(self, joints_xy, vis):
        """``_target_generator`` (:122-158) for (B,J,2) input-pixel joints, (B,J) visibility -> target (B,J,h,w), weight (B,J,1,1)."""
        dev = _device()
        j = vh.upload(np.ascontiguousarray(joints_xy, np.float32), dev)
        v = vh.upload(np.ascontiguousarray(vis, np.float32), dev)
        return vh.gaussian_targets(j, v, tuple(int(s) for s in self._heatmap_size), tuple(int(s) for s in self._input_size), float(self._sigma))

    # ------------------------------------------------------------------ reference per-item interface
    def test_transform(self, img, bbox):
        xmin, ymin, xmax, ymax = bbox
        center, scale = _box_to_center_scale(xmin, ymin, xmax - xmin, ymax - ymin, self._aspect_ratio)
        scale = scale * 1.0
        arena = FrameArena([img])
        crops, _ = self.crop_batch(arena, [0], center[None], scale[None], 0.0)
        return crops[0], torch.tensor(_center_scale_to_box(center, scale))

    def _target_generator(self, joints_3d, num_joints):
        target, weight = self.targets_batch(joints_3d[None, :, 0:2, 0], joints_3d[None, :, 0, 1])
        return target[0], weight[0]

    # joint counts whose leading (body) joints weigh double in the integral loss (:164-169): whole-body 136 / 133, face 68
    _DOUBLE_WEIGHT_HEAD = {136: 26, 133: 23, 68: 26}

    def _integral_target_generator(self, joints_3d, num_joints, patch_height, patch_width):
        """Soft-arg-max regression targets (:160-177): per joint (x / W - 0.5, y / H - 0.5), interleaved; the weights are the
        visibility flag on both coordinates, doubled for the leading joints of the whole-body layouts."""
        j = np.asarray(joints_3d)[:num_joints]
        boost = np.ones(num_joints, np.float32)
        boost[:self._DOUBLE_WEIGHT_HEAD.get(num_joints, 0)] = 2
        weight = np.repeat((j[:, 0, 1] * boost).astype(np.float32), 2)
        target = np.stack([j[:, 0, 0] / patch_width - 0.5, j[:, 1, 0] / patch_height - 0.5], 1).astype(np.float32)
        return target.reshape(-1), weight

    def _draw(self, label):
        """Host half of ``__call__`` (:179-229): centre/scale of the box and, in train mode, the augmentation draws in the
        reference's order (half-body, scale, rotation, flip) so a seeded run selects the same augmentations.
        -> center, scale (float32 (2,)), rotation in degrees, mirror flag, joints (J,3,2) (flipped when mirrored)."""
        bbox = list(label["bbox"])
        xmin, ymin, xmax, ymax = bbox
        center, scale = _box_to_center_scale(xmin, ymin, xmax - xmin, ymax - ymin, self._aspect_ratio)
        if self._add_dpg and self._train:
            raise NotImplementedError("DPG box jitter is broken in the reference itself (simple_transform.py:184-185 reads imgwidth "
                                      "before it is assigned) and no shipped config enables it")
        imgwidth = label["width"]
        gt_joints = label["joints_3d"]
        self.num_joints = gt_joints.shape[0]
        joints_vis = np.zeros((self.num_joints, 1), dtype=np.float32)
        joints_vis[:, 0] = gt_joints[:, 0, 1]
        if self._train and (np.sum(joints_vis[:, 0]) > self.num_joints_half_body and np.random.rand() < self.prob_half_body):
            c_half_body, s_half_body = self.half_body_transform(gt_joints[:, :, 0], joints_vis)
            if c_half_body is not None and s_half_body is not None:
                center, scale = c_half_body, s_half_body
        if self._train:
            sf = self._scale_factor
            scale = (scale * np.float32(np.clip(np.random.randn() * sf + 1, 1 - sf, 1 + sf))).astype(np.float32)
        else:
            scale = scale * 1.0
        if self._train:
            rf = self._rot
            r = np.clip(np.random.randn() * rf, -rf * 2, rf * 2) if random.random() <= 0.6 else 0
        else:
            r = 0
        joints = gt_joints
        mirror = False
        if random.random() > 0.5 and self._train:
            mirror = True                                   # the kernel reads the frame right-to-left instead of copying it
            joints = flip_joints_3d(joints, imgwidth, self._joint_pairs)
            center[0] = imgwidth - center[0] - 1
        return center, scale, r, mirror, joints

    def call_batch(self, arena: FrameArena, frame_index, labels):
        """``__call__`` (:179-251) for a batch: per-item draws on the host, then ONE warp launch and ONE target launch.
        -> crops (B,3,H,W), targets, target weights, boxes (B,4) float32 (host tensor)."""
        if not self._train:                                                             # no draws: the arithmetic of `_draw` on arrays
            return self.eval_batch(arena, frame_index, np.array([lb["bbox"] for lb in labels], np.float64).reshape(-1, 4),
                                   np.stack([lb["joints_3d"] for lb in labels]))
        draws = [self._draw(lb) for lb in labels]
        centers = np.stack([d[0] for d in draws]).astype(np.float32).reshape(-1, 2)
        scales = np.stack([d[1] for d in draws]).astype(np.float32).reshape(-1, 2)
        rots = np.array([d[2] for d in draws], np.float64)
        crops, trans = self.crop_batch(arena, frame_index, centers, scales, rots, mirror=[d[3] for d in draws])
        joints = np.stack([d[4] for d in draws]).astype(np.float32)                     # (B,J,3,2), a copy
        return self._finish_batch(crops, trans, joints, centers, scales)

    def eval_batch(self, arena: FrameArena, frame_index, boxes_xyxy, joints_3d):
        """``call_batch`` of an evaluation-mode transform from ARRAYS (boxes (B,4) xyxy float64, joints (B,J,3,2)): without augmentation
        `_draw` is the box -> centre / scale arithmetic alone (:179-229 with `train` off), so a batch needs no per-item host work."""
        if self._train:
            raise ValueError("eval_batch: this transform draws augmentations (train mode); use call_batch")
        centers, scales = box_to_center_scale_batch(boxes_xyxy, self._aspect_ratio)
        crops, trans = self.crop_batch(arena, frame_index, centers, scales, 0.0)
        self.num_joints = int(np.shape(joints_3d)[1])
        return self._finish_batch(crops, trans, np.array(joints_3d, np.float32), centers, scales)

    def _finish_batch(self, crops, trans, joints, centers, scales):
        """Joints through the crop's affine map, targets / weights on the device, the corrected boxes (:231-251)."""
        inp_h, inp_w = self._input_size
        xy1 = np.concatenate([joints[:, :, 0:2, 0].astype(np.float64), np.ones(joints.shape[:2] + (1,))], 2)
        moved = np.einsum("bik,bjk->bji", trans, xy1)                                   # affine_transform (:789-792) per joint
        vis = joints[:, :, 0, 1] > 0.0
        joints[:, :, 0:2, 0] = np.where(vis[..., None], moved, joints[:, :, 0:2, 0])
        if self._loss_type == "MSELoss":
            target, weight = self.targets_batch(joints[:, :, 0:2, 0], joints[:, :, 0, 1])
        elif "JointRegression" in self._loss_type:
            tw = [self._integral_target_generator(j, j.shape[0], inp_h, inp_w) for j in joints]
            target = torch.from_numpy(np.stack([t for t, _ in tw]))
            weight = torch.from_numpy(np.stack([w for _, w in tw]))
        else:
            raise NotImplementedError(self._loss_type)
        boxes = torch.from_numpy(center_scale_to_box_batch(centers, scales).astype(np.float32))
        return crops, target, weight, boxes

    def __call__(self, img, label):
        imgwidth, imght = label["width"], label["height"]
        assert imgwidth == img.shape[1] and imght == img.shape[0]
        assert img.shape[2] == 3
        crops, target, weight, boxes = self.call_batch(FrameArena([img]), [0], [label])
        return crops[0], target[0], weight[0], torch.Tensor(boxes[0].tolist())

    def half_body_transform(self, joints, joints_vis):
        """Centre / scale of a box around the visible upper-body or lower-body joints (:253-304).  One normal deviate decides
        for the upper body (drawn whatever the joint counts are, like the reference's short-circuit order); a half needs more
        than two visible joints to be eligible, and fewer than two selected joints means "no half-body crop"."""
        ids = np.arange(self.num_joints)
        seen = np.asarray(joints_vis)[:self.num_joints, 0] > 0
        upper = seen & np.isin(ids, np.asarray(self.upper_body_ids))
        lower = seen & ~np.isin(ids, np.asarray(self.upper_body_ids))
        prefer_upper = np.random.randn() < 0.5
        if prefer_upper and upper.sum() > 2:
            pick = upper
        else:
            pick = lower if lower.sum() > 2 else upper
        if pick.sum() < 2:
            return None, None
        pts = np.asarray(joints, dtype=np.float32)[:self.num_joints][pick]         # visible joints of the chosen half, in joint order
        center = pts.mean(axis=0)[:2]
        w, h = (pts.max(axis=0) - pts.min(axis=0))[:2]
        ar = self._aspect_ratio
        if w > ar * h:                                                             # grow the short side to the crop's aspect ratio
            h = w * 1.0 / ar
        elif w < ar * h:
            w = h * ar
        return center, np.array([w * 1.0 / self.pixel_std, h * 1.0 / self.pixel_std], dtype=np.float32) * 1.5
