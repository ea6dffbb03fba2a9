"""PoseTrack21 / JRDB-Pose video datasets from COCO-format annotation json (reference: alphapose/datasets/custom.py:24-176,
posetrack21.py:12-129, jrdb2022.py).

The annotation semantics are the reference's (``_load_jsons`` / ``_check_load_keypoints``): boxes converted xywh -> xyxy
and clipped to the frame, persons without a positive box / key-points / visible joint dropped, one item per person, items
sorted by ``id`` = last two (PoseTrack) or three (JRDB) digits of the annotation id + image id, ``track_id`` as the reference
builds it.  Only the plumbing differs: the json is indexed with a few dict lookups instead of pycocotools, frames are decoded
on demand with Pillow (the reference uses ``cv2.imread`` + BGR->RGB; both sit on libjpeg, neither is part of the GPU path)
and the crops / targets of a whole batch come from the device (``FrameVideo``).
Config keys as in the reference's yaml: ``ROOT``, ``IMG_PREFIX``, ``ANN``, optional ``AUG``; ``PRESET`` is injected by
``builder.build_dataset``.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import numpy as np

from alphapose.models.builder import DATASET
from alphapose.utils.bbox import bbox_clip_xyxy, bbox_xywh_to_xyxy
from alphapose.utils.presets.simple_transform import FrameArena

from .frame_video import FrameVideo


class CocoIndex:
    """The four pycocotools queries the loaders need (getCatIds/loadCats, getImgIds/loadImgs, getAnnIds/loadAnns)."""

    def __init__(self, path):
        with open(path) as f:
            data = json.load(f)
        self.cats = list(data.get("categories", []))
        self.imgs = {im["id"]: im for im in data.get("images", [])}
        self.anns_of = {}
        for a in data.get("annotations", []):
            self.anns_of.setdefault(a["image_id"], []).append(a)

    def category_names(self):
        return [c["name"] for c in self.cats]

    def images_sorted(self):
        return [self.imgs[i] for i in sorted(self.imgs)]


def _read_rgb(path):
    try:
        from PIL import Image
    except ImportError as e:                                     # pragma: no cover
        raise RuntimeError("decoding video frames needs Pillow (the reference uses cv2.imread)") from e
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


class _CocoVideo(FrameVideo):
    CLASSES = ["person"]
    num_joints = 17
    ID_DIGITS = 2                     # digits of the annotation id that lead the sort key (posetrack21.py:103)
    STRICT_BOX = True                 # posetrack21.py:84 drops xmax <= xmin; jrdb2022.py only xmax < xmin
    FRAME_CACHE = 64                  # decoded frames kept on the host (an id-sorted stream revisits a frame for every person in it)

    def __init__(self, train=True, dpg=False, skip_empty=True, lazy_import=False, get_prenext=False, **cfg):
        if dpg:
            raise NotImplementedError("DPG augmentation is broken in the reference itself (simple_transform.py:184-185)")
        self._root, self._img_prefix = cfg["ROOT"], cfg.get("IMG_PREFIX", "")
        self._ann_file = os.path.join(self._root, cfg["ANN"])
        self._skip_empty = skip_empty
        db = CocoIndex(self._ann_file)
        assert db.category_names() == self.CLASSES, "Incompatible category names with " + type(self).__name__
        anns = []
        for count, frame in enumerate(db.images_sorted()):
            path = self._image_path(frame)
            if not os.path.exists(path):
                raise IOError(f"Image: {path} not exists.")
            persons = self._persons(db, frame)
            for person in persons:
                person["frame"] = path
                person["img_id"] = frame.get("image_id", frame["id"])
                person.setdefault("id", len(anns))               # image datasets keep file order
                anns.append(person)
            if persons and self._stop_after(count + 1):          # (the reference checks its image counter only after a labelled image)
                break
        super().__init__(frames=[], annotations=anns, train=train, get_prenext=get_prenext, PRESET=cfg.get("PRESET"), AUG=cfg.get("AUG"))
        self._items = [{"path": a["frame"], "img_id": a["img_id"], "ann_id": a.get("ann_id"), "id": a["id"], "track_id": a.get("track_id"),
                        "keypoint": a.get("keypoint")} for a in self._labels]
        self._decoded = OrderedDict()

    def _image_path(self, frame):
        return os.path.join(self._root, frame["file_name"])      # posetrack21.py:53-54

    def _stop_after(self, n_images):
        return False

    def _track_id(self, frame, obj):
        return str(frame["vid_id"]) + str(obj["track_id"])       # posetrack21.py:105

    def _persons(self, db, frame):
        """``_check_load_keypoints`` (posetrack21.py:77-129)."""
        width, height = int(frame["width"]), int(frame["height"])
        out = []
        for obj in db.anns_of.get(frame["image_id"], db.anns_of.get(frame["id"], [])):
            if "bbox" not in obj:
                continue
            xmin, ymin, xmax, ymax = bbox_clip_xyxy(bbox_xywh_to_xyxy(obj["bbox"]), width, height)
            if (xmax <= xmin or ymax <= ymin) if self.STRICT_BOX else (xmax < xmin or ymax < ymin):
                continue
            if max(obj["keypoints"]) == 0:
                continue
            joints_3d = np.zeros((self.num_joints, 3, 2), dtype=np.float32)
            kp = obj["keypoints"]
            for i in range(self.num_joints):
                joints_3d[i, 0, 0], joints_3d[i, 1, 0] = kp[i * 3 + 0], kp[i * 3 + 1]
                joints_3d[i, :2, 1] = min(1, kp[i * 3 + 2])
            if np.sum(joints_3d[:, 0, 1]) < 1:
                continue
            ann_id = int(obj["id"])
            out.append({"bbox": (float(xmin), float(ymin), float(xmax), float(ymax)), "width": width, "height": height, "joints_3d": joints_3d,
                        "keypoint": list(kp), "id": int(str(ann_id)[-self.ID_DIGITS:] + str(frame["image_id"])), "ann_id": ann_id,
                        "track_id": self._track_id(frame, obj)})
        if not out and not self._skip_empty:                     # dummy invalid label (posetrack21.py:117-128)
            out.append({"bbox": (-1.0, -1.0, 0.0, 0.0), "width": width, "height": height, "joints_3d": np.zeros((self.num_joints, 3, 2), np.float32),
                        "keypoint": [0.0] * (self.num_joints * 3), "id": -1, "ann_id": -1, "track_id": -1})
        return out

    def _frames_for(self, keys):
        uniq = list(OrderedDict.fromkeys(keys))
        frames = []
        for path in uniq:
            img = self._decoded.get(path)
            if img is None:
                img = _read_rgb(path)
                self._decoded[path] = img
                while len(self._decoded) > self.FRAME_CACHE:
                    self._decoded.popitem(last=False)
            else:
                self._decoded.move_to_end(path)
            frames.append(img)
        return FrameArena(frames), {path: k for k, path in enumerate(uniq)}


@DATASET.register_module
class Posetrack21(_CocoVideo):
    """PoseTrack21 (posetrack21.py:12-24)."""
    EVAL_JOINTS = list(range(17))
    joint_pairs = [[5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]


@DATASET.register_module
class JRDB2022(_CocoVideo):
    """JRDB-Pose (jrdb2022.py:12-25)."""
    EVAL_JOINTS = list(range(17))
    joint_pairs = [[1, 2], [0, 4], [3, 4], [8, 10], [5, 7], [10, 13], [14, 16], [4, 5], [7, 12], [4, 8], [3, 6], [13, 15], [11, 14], [6, 9], [8, 11]]
    ID_DIGITS = 3
    STRICT_BOX = False

    def _track_id(self, frame, obj):
        return obj["track_id"]


class _CocoImages(_CocoVideo):
    """Image-only key-point datasets (pre-training): items are ``CustomDataset.__getitem__``'s 5-tuple
    (img (3,H,W), label, label_mask, img_id, bbox) (custom.py:96-113)."""
    REQUIRE_AREA = False

    def __init__(self, train=True, dpg=False, skip_empty=True, lazy_import=False, get_prenext=False, **cfg):
        super().__init__(train=train, dpg=dpg, skip_empty=skip_empty, lazy_import=lazy_import, get_prenext=False, **cfg)
        self.ID_SORTED_STREAM = False

    def _persons(self, db, frame):
        """``_check_load_keypoints`` of mscoco.py:60-115 / mpii.py:62-110 (``_check_centers`` is off in the reference)."""
        width, height = frame["width"], frame["height"]
        out = []
        for obj in db.anns_of.get(frame["id"], []):
            if obj.get("iscrowd", 0):
                continue
            if max(obj["keypoints"]) == 0:
                continue
            xmin, ymin, xmax, ymax = bbox_clip_xyxy(bbox_xywh_to_xyxy(obj["bbox"]), width, height)
            if (self.REQUIRE_AREA and obj["area"] <= 0) or xmax <= xmin or ymax <= ymin:
                continue
            if obj["num_keypoints"] == 0:
                continue
            joints_3d = np.zeros((self.num_joints, 3, 2), dtype=np.float32)
            kp = obj["keypoints"]
            for i in range(self.num_joints):
                joints_3d[i, 0, 0], joints_3d[i, 1, 0] = kp[i * 3 + 0], kp[i * 3 + 1]
                joints_3d[i, :2, 1] = min(1, kp[i * 3 + 2])
            if np.sum(joints_3d[:, 0, 1]) < 1:
                continue
            out.append({"bbox": (float(xmin), float(ymin), float(xmax), float(ymax)), "width": width, "height": height, "joints_3d": joints_3d})
        if not out and not self._skip_empty:
            out.append({"bbox": (-1.0, -1.0, 0.0, 0.0), "width": width, "height": height, "joints_3d": np.zeros((self.num_joints, 3, 2), np.float32)})
        return out

    collated = None                                          # (5-tuple items: no video columns; loaders go through __getitems__)

    def __getitems__(self, idxs):
        idxs = [int(i) for i in idxs]
        labels = [dict(self._labels[i]) for i in idxs]
        arena, where = self._frames_for([lb["frame"] for lb in labels])
        img, target, weight, boxes = self.transformation.call_batch(arena, [where[lb["frame"]] for lb in labels], labels)
        ids = [int(os.path.splitext(os.path.basename(lb["frame"]))[0]) for lb in labels]      # custom.py:103
        return [(img[k], target[k], weight[k], ids[k], boxes[k]) for k in range(len(idxs))]

    @staticmethod
    def my_collate_fn(batch):
        import torch
        cols = list(zip(*batch))
        return torch.stack(cols[0]), torch.stack(cols[1]), torch.stack(cols[2]), list(cols[3]), torch.stack(cols[4])


@DATASET.register_module
class Mscoco(_CocoImages):
    """COCO person key-points (mscoco.py:9-25).  ``SHORTEN`` is the reference's own switch: it stops after 30 images."""
    EVAL_JOINTS = list(range(17))
    joint_pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    SHORTEN = True
    REQUIRE_AREA = True

    def _image_path(self, frame):
        dirname, filename = frame["coco_url"].split("/")[-2:]    # mscoco.py:41-42
        return os.path.join(self._root, dirname, filename)

    def _stop_after(self, n_images):
        return self.SHORTEN and n_images >= 30                   # mscoco.py:53-55


@DATASET.register_module
class Mpii(_CocoImages):
    """MPII human pose, 16 joints (mpii.py:13-34)."""
    num_joints = 16
    EVAL_JOINTS = list(range(16))
    joint_pairs = [[0, 5], [1, 4], [2, 3], [10, 15], [11, 14], [12, 13]]

    def _image_path(self, frame):
        return os.path.join(self._root, self._img_prefix, frame["file_name"])                 # mpii.py:47-48
