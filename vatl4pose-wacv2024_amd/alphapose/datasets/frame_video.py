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

    def _columns(self):
        """The annotation fields of every item as arrays (built once: the labels do not change after construction) — what an evaluation
        batch slices instead of touching 256 dicts: boxes xyxy float64, joints (N,J,3,2), GT key-points and annotation boxes as float32,
        image / annotation ids, frame keys, same-track flags of the id-neighbours.  numpy arrays, sliced by numpy: a torch CPU indexing op
        here starts torch's intra-op thread pool (one thread per host core, 256 on the MI355X boxes), whose spinning workers delayed the HIP
        runtime's own threads — launches of the scoring loop then blocked for a whole device backlog (rounds of 90 - 300 ms instead of 71)."""
        c = self.__dict__.get("_cols")
        if c is None or len(c["frame"]) != len(self._labels):
            lb = self._labels
            n = len(lb)
            track = [a["track_id"] for a in lb]
            same = np.array([track[i] == track[i + 1] for i in range(n - 1)], bool) if n > 1 else np.zeros(0, bool)
            c = self._cols = {
                "bbox": np.array([a["bbox"] for a in lb], np.float64).reshape(n, 4),
                "joints": np.stack([a["joints_3d"] for a in lb]).astype(np.float32) if n else np.zeros((0, 17, 3, 2), np.float32),
                "kp": np.array([a["keypoint"] for a in lb], np.float64).astype(np.float32).reshape(n, -1),
                "bbox_ann": np.array([a["bbox"] for a in lb], np.float64).astype(np.float32).reshape(n, 4),
                "img_id": [a["img_id"] for a in lb], "ann_id": [a["ann_id"] for a in lb], "frame": [a["frame"] for a in lb],
                "prev": np.concatenate([[False], same]), "next": np.concatenate([same, [False]])}
        return c

    def invalidate_columns(self):
        """Call after editing an annotation dict of ``_labels`` IN PLACE (a corrected box, a pseudo-label): evaluation batches read the
        cached columns, training batches and `annotation_of` read the dicts — without this call the two would disagree.  (The reference's
        datasets build their labels once and never edit them, posetrack21.py:40-129; appending / removing items is noticed by itself.)"""
        self.__dict__.pop("_cols", None)

    def collated(self, idxs):
        """The batch `my_collate_fn(__getitems__(idxs))` makes, without the per-item detour: the 11 columns directly.  In evaluation mode
        nothing is done per item on the host (array slices of `_columns`, one warp launch, one target launch)."""
        idxs = np.asarray([int(i) for i in idxs], np.int64)
        n = len(idxs)
        c = self._columns()
        is_prev = (c["prev"][idxs] if self.get_prenext else np.zeros(n, bool)).tolist()
        is_next = (c["next"][idxs] if self.get_prenext else np.zeros(n, bool)).tolist()
        frames = [c["frame"][i] for i in idxs]
        keys = list(frames)
        if self.emit_neighbour_crops:
            keys += [c["frame"][i - 1] for i, f in zip(idxs, is_prev) if f] + [c["frame"][i + 1] for i, f in zip(idxs, is_next) if f]
        arena, where = self._frames_for(keys)
        at = (lambda k: k) if where is None else (lambda k: where[k])
        st = self.transformation
        if self._train:
            labels = [dict(self._labels[i]) for i in idxs]      # shallow: the transform copies the joints it moves, nothing else is written
            for lb in labels:
                h, w = arena.hw[at(lb["frame"])]
                lb.setdefault("width", int(w)); lb.setdefault("height", int(h))
            cur, target, weight, boxes = st.call_batch(arena, [at(f) for f in frames], labels)
        else:
            cur, target, weight, boxes = st.eval_batch(arena, [at(f) for f in frames], c["bbox"][idxs], c["joints"][idxs])
        slots = 3 if self.emit_neighbour_crops else 1
        stacked = torch.zeros((n, slots) + tuple(cur.shape[1:]), device=cur.device) if slots == 3 else cur[:, None]
        if slots == 3:
            stacked[:, 0] = cur
        for slot, flags, step in ((1, is_prev, -1), (2, is_next, +1)):               # test_transform of the neighbour (:154-178)
            rows = [k for k in range(n) if flags[k]] if slots == 3 else []
            if rows:
                nb = idxs[rows] + step
                crops, _ = st.test_transform_batch(arena, [at(c["frame"][j]) for j in nb], c["bbox"][nb])
                stacked[vh.upload(np.asarray(rows, np.int64), cur.device), slot] = crops
        return (idxs.tolist(), stacked, target, weight, torch.from_numpy(c["kp"][idxs]), [c["img_id"][i] for i in idxs], [c["ann_id"][i] for i in idxs],
                boxes, torch.from_numpy(c["bbox_ann"][idxs]), is_prev, is_next)

    def __getitems__(self, idxs):
        cols = self.collated(idxs)
        return [tuple(col[k] for col in cols) for k in range(len(cols[0]))]

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
