"""Video dataset over in-memory uint8 frames, cropping on MI355X (item contract of posetrack21.py:131-224).

The reference's ``Posetrack21`` / ``JRDB2022`` read COCO-style json and decode jpeg files on DataLoader workers
(posetrack21.py:59-129, ``cv2.imread`` :141); that file I/O stays outside this build.  ``FrameVideo`` takes what those
loaders produce — decoded RGB frames and one annotation dict per person (``bbox`` xyxy, ``joints_3d`` (J,3,2),
``keypoint`` (3J,), ``ann_id``, ``img_id``, ``track_id``, ``frame`` = index into ``frames``, ``id`` = sort key) — and yields
the same 11-tuple:
  (idx, stacked_inp (3,3,H,W) [current, prev, next], label (J,h,w), label_mask (J,1,1), GTkpt (3J,), img_id, ann_id,
   bbox_crop (4,), bbox_ann (4,), isPrev, isNext)
with the crops, targets and masks made by ``SimpleTransform`` on the device: a whole DataLoader batch is ONE warp launch
(``__getitems__``), the frames live in HBM once (``FrameArena``).  Items are id-sorted (posetrack21.py:71-72); prev/next
exist when the id-adjacent item belongs to the same track (:148-178).  In eval mode the prev/next crops are exactly the
neighbours' current crops (same frame, same box, no augmentation), which ``ID_SORTED_STREAM`` declares so the scoring
loop runs one forward per item.
"""
from __future__ import annotations


import numpy as np
import torch
from torch.utils.data import Dataset

import vatl_hip as vh

from alphapose.models.builder import DATASET
from alphapose.utils.presets.simple_transform import FrameArena, SimpleTransform


@DATASET.register_module
class FrameVideo(Dataset):
    EVAL_JOINTS = list(range(17))
    DEVICE_ITEMS = True                    # items hold device tensors: loaders must not pin them
    joint_pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    upper_body_ids = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10)       # custom.py:87-88
    lower_body_ids = (11, 12, 13, 14, 15, 16)

    def __init__(self, frames=None, annotations=None, train=False, get_prenext=True, PRESET=None, AUG=None, LOSS_TYPE="MSELoss", **_):
        if frames is None or annotations is None:
            raise ValueError("FrameVideo needs decoded frames and their annotations (see the module docstring)")
        preset = PRESET or {}
        self._train, self.get_prenext = bool(train), bool(get_prenext)
        self.ID_SORTED_STREAM = not self._train
        aug = AUG or {}
        if self._train:                                          # custom.py:68-77
            self.num_joints_half_body = aug.get("NUM_JOINTS_HALF_BODY", 8)
            self.prob_half_body = aug.get("PROB_HALF_BODY", -1)
            scale_factor, rot = aug.get("SCALE_FACTOR", 0), aug.get("ROT_FACTOR", 0)
        else:
            self.num_joints_half_body, self.prob_half_body, scale_factor, rot = -1, -1, 0, 0
        self._input_size = list(preset.get("IMAGE_SIZE", [256, 192]))
        self._output_size = list(preset.get("HEATMAP_SIZE", [64, 48]))
        self.transformation = SimpleTransform(self, scale_factor=scale_factor, input_size=self._input_size, output_size=self._output_size,
                                              rot=rot, sigma=preset.get("SIGMA", 2), train=self._train, add_dpg=False,
                                              loss_type=preset.get("LOSS_TYPE", LOSS_TYPE))
        self._labels = sorted((dict(a) for a in annotations), key=lambda a: a["id"])     # posetrack21.py:71-72
        # A consumer that takes the neighbours' heat-maps from the id-sorted stream itself (ActiveLearning with
        # ID_SORTED_STREAM) switches the prev / next crops off: items then carry a (1,3,H,W) stack and the flags only.
        self.emit_neighbour_crops = True
        self._frames = frames
        self._arena = None

    def __len__(self):
        return len(self._labels)

    @property
    def arena(self) -> FrameArena:
        if self._arena is None:                                  # frames go to HBM once, on first use
            self._arena = FrameArena(self._frames)
        return self._arena

    def _frames_for(self, keys):
        """-> (arena, {frame key: index in that arena}) holding at least the frames ``keys``.  In-memory videos keep every
        frame resident; file-backed subclasses decode and upload what one batch needs."""
        return self.arena, None

    def _neighbour(self, i, step):
        j = i + step
        return 0 <= j < len(self._labels) and self._labels[j]["track_id"] == self._labels[i]["track_id"]

    def __getitems__(self, idxs):
        idxs = [int(i) for i in idxs]
        labels = [dict(self._labels[i]) for i in idxs]          # shallow: the transform copies the joints it moves, nothing else is written
        is_prev = [self.get_prenext and self._neighbour(i, -1) for i in idxs]
        is_next = [self.get_prenext and self._neighbour(i, +1) for i in idxs]
        keys = [lb["frame"] for lb in labels]
        if self.emit_neighbour_crops:
            keys += [self._labels[i - 1]["frame"] for i, f in zip(idxs, is_prev) if f] + [self._labels[i + 1]["frame"] for i, f in zip(idxs, is_next) if f]
        arena, where = self._frames_for(keys)
        at = (lambda k: k) if where is None else (lambda k: where[k])
        for lb in labels:
            h, w = arena.hw[at(lb["frame"])]
            lb.setdefault("width", int(w)); lb.setdefault("height", int(h))
        st = self.transformation
        cur, target, weight, boxes = st.call_batch(arena, [at(lb["frame"]) for lb in labels], labels)
        n = len(idxs)
        slots = 3 if self.emit_neighbour_crops else 1
        stacked = torch.zeros((n, slots) + tuple(cur.shape[1:]), device=cur.device) if slots == 3 else cur[:, None]
        if slots == 3:
            stacked[:, 0] = cur
        for slot, flags, step in ((1, is_prev, -1), (2, is_next, +1)):               # test_transform of the neighbour (:154-178)
            rows = [k for k in range(n) if flags[k]] if slots == 3 else []
            if rows:
                nb = [self._labels[idxs[k] + step] for k in rows]
                crops, _ = st.test_transform_batch(arena, [at(a["frame"]) for a in nb], np.array([a["bbox"] for a in nb], np.float64))
                stacked[vh.upload(np.asarray(rows, np.int64), cur.device), slot] = crops
        out = []
        for k, (i, lb) in enumerate(zip(idxs, labels)):
            out.append((i, stacked[k], target[k], weight[k], torch.tensor(lb["keypoint"], dtype=torch.float32), lb["img_id"], lb["ann_id"],
                        boxes[k], torch.Tensor(list(lb["bbox"])), bool(is_prev[k]), bool(is_next[k])))
        return out

    def __getitem__(self, i):
        return self.__getitems__([i])[0]

    def annotation_of(self, i):
        """(GT key-points (3J,), annotation box xyxy) of item i — what the auto-encoder refit reads — without touching pixels."""
        lb = self._labels[int(i)]
        return np.asarray(lb["keypoint"], np.float32), np.asarray(lb["bbox"], np.float32)

    @staticmethod
    def my_collate_fn(batch):
        cols = list(zip(*batch))
        return (list(cols[0]), torch.stack(cols[1]), torch.stack(cols[2]), torch.stack(cols[3]), torch.stack(cols[4]),
                list(cols[5]), list(cols[6]), torch.stack(cols[7]), torch.stack(cols[8]), list(cols[9]), list(cols[10]))
